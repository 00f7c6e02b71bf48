#!/bin/bash
# round 3: whole -m gpu suite, MMoE line, then the PMC passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 ) | tail -40 | tee $O/gpu_suite.txt
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| dom', (r.get('kernel') or '')[:60], r.get('us_per_step'), r.get('frac'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), p.get('error'))
for f in r.get('families', []): print('   ', f['family'], round(f['us_per_step'],1), round(f['share'],3), f['launches_per_step'])
for k in r.get('kernels', [])[:12]: print('      ', round(k['us_per_step'],1), k['launches_per_step'], k['kernel'][:80])
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 1200 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --cpu_seconds 2
bash tools/gpu_round3_pmc.sh 2>&1 | tail -80
