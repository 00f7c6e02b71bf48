#!/bin/bash
# BatchNorm on the moving statistics (MMoE / DBMTL experts): kernel + model parity, then the MMoE bench line and its kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02frozen; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
( time timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short --timeout 120 -k "bn_act or frozen or bn_matches" 2>&1 | cut -c1-250 | tail -15 ) > $O/pytest_kernels.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E |^real" $O/pytest_kernels.log | head -12
( time timeout 400 python -m pytest tests/test_models_gpu.py tests/test_golden_models.py tests/test_grad_clip.py tests/test_embedding_parallel_gpu.py -m gpu -q --tb=short --timeout 150 -k "mmoe or dbmtl or golden or neighbouring" 2>&1 | cut -c1-250 | tail -25 ) > $O/pytest_models.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E |^real|ms/step" $O/pytest_models.log | head -20
( time timeout 300 python bench.py --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128 ) > $O/mmoe25m.out 2>&1
grep '^{' $O/mmoe25m.out | tail -1 > $O/mmoe25m.json; python -c "
import json; d=json.load(open('$O/mmoe25m.json')); s=d.get('steady_state') or {}; r=d['roofline']; print('mmoe25m', round(d['ms_per_step'],4), round(d['value']), 'steady', round(s.get('ms_per_step_mean',0),4), r.get('kernel'), round(r['frac'],3))"; grep -E "^real|Error|Traceback" $O/mmoe25m.out | head -3
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --config $GRAFT_REPO_ROOT/configs/mmoe_taobao_4task_d64_25m.config --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 64 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_mmoe25m.csv --steps 284 | tail -45 > $O/stats.txt; rm -rf $O/prof; head -30 $O/stats.txt | cut -c1-160
