#!/bin/bash
# round 3, fourth GPU call: the whole -m gpu suite (no -x), the default bench line, the 128-tile weight-gradient kernel A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q --durations=12 2>&1 ) | tail -60 | tee $O/gpu_suite.txt
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}; c=d.get('cpu_baseline') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| dom', (r.get('kernel') or '')[:60], r.get('us_per_step'), r.get('frac'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), p.get('error'), '| cpu', c.get('value'), c.get('cores'))
for f in r.get('families', []): print('   ', f['family'], round(f['us_per_step'],1), round(f['share'],3), f['launches_per_step'])
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 1200 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
run default
ER_GEMM_TN128=1 run tn128 --no_cpu_baseline --steady_steps 256
run din10m --config configs/din_taobao_10m.config --no_cpu_baseline --steady_steps 128 --precondition 128
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128
