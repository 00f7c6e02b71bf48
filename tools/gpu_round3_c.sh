#!/bin/bash
# round 3, third GPU call: the whole -m gpu suite, then the bench lines of BASELINE configs 2-5 WITH the full-size oracle
# comparison (parity_full_size) and the CPU baseline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 ) | tail -45 | tee $O/gpu_suite.txt
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
run dcnv2_f32 --config configs/dcn_v2_criteo.config --steady_steps 256 --precondition 256 --cpu_seconds 4
run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 --steady_steps 256 --precondition 256 --cpu_seconds 4
run din10m --config configs/din_taobao_10m.config --steady_steps 128 --precondition 128 --cpu_seconds 4
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --cpu_seconds 4
