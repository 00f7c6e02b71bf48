#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02cin; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -q --tb=short --timeout 150 -k "cin or xdeepfm" 2>&1 | grep -E "passed|failed|^E  |FAILED" | head -12
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
prof() { name=$1; shift; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$name -o step -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 200 --warmup 20 --steady_steps 0 --precondition 64 "$@" > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1; cd $GRAFT_REPO_ROOT; DB=$(find $O/prof_$name -name "*.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_$name.csv --steps 284 | tail -45 > $O/stats_$name.txt; rm -rf $O/prof_$name; tail -1 $O/stats_$name.txt; }
prof xdeepfm --config $GRAFT_REPO_ROOT/configs/xdeepfm_taobao.config
