#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bb; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
( time timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout 200 2>&1 | grep -v "^$" | cut -c1-300 | tail -30 ) > $O/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E |^real" $O/pytest.log | head -20
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| loss', d.get('final_loss'))"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; timeout 300 python bench.py --no_cpu_baseline "$@" > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
run xdeepfm --config configs/xdeepfm_taobao.config --steady_steps 128 --precondition 128
