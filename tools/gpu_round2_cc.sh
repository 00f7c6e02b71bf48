#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02cc; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'ms/step | steady', round(s.get('ms_per_step_mean',0),4))"; }
run() { echo "--- $*" | tee -a $O/lines.log; env $ENVV timeout 300 python bench.py --no_cpu_baseline --steps 150 "$@" 2>&1 | grep '^{' | tail -1 | line | tee -a $O/lines.log; }
for r in 1024 2048 4096 8192; do ENVV="ER_WGRAD_SPLIT_ROWS=$r" run --config configs/din_taobao_10m.config --steady_steps 0 --precondition 64; done
for t in 1 2 4; do ENVV="ER_BN_TILES_MID=$t" run --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 0 --precondition 64; done
for i in 1 2; do ( time timeout 600 python -m pytest tests -m gpu -q --tb=line --timeout 200 2>&1 | tail -3 ) 2>&1 | grep -E "passed|failed|real" | tee -a $O/suite.log; done
