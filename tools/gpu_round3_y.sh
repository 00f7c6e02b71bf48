#!/bin/bash
# split-K heuristics of the grouped weight-gradient launch, re-measured on exact grids
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step |', ' | '.join('%s %.1f' % (k['kernel'][14:50], k['us_per_step']) for k in (r.get('kernels') or [])[:40] if 'grouped' in k['kernel'] and ('false, false' in k['kernel'] or 'reduce' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
D="--config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
C="--config configs/dcn_v2_criteo.config --steady_steps 128 --precondition 128 --no_cpu_baseline --parity_steps 0"
for combo in "512 2048" "1024 2048" "512 1024" "256 2048"; do
  set -- $combo
  export ER_WGRAD_TARGET_BLOCKS=$1 ER_WGRAD_SPLIT_ROWS=$2
  run mmoe_tb$1_rows$2 $M
  run dcnv2_tb$1_rows$2 $C
  run din_tb$1_rows$2 $D
done
