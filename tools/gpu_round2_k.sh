#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
timeout 1200 python -m pytest tests/test_embedding_parallel_gpu.py tests/test_multi_rank_oracle_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -20
timeout 300 python tools/trace_step_ops.py > $O/step_ops.txt 2>&1; grep -c "^lib\|^aten" $O/step_ops.txt; grep "^aten" $O/step_ops.txt
for ov in 1 0; do
cd /tmp && EASYREC_AMD_OVERLAP_FLUSH=$ov timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof$ov -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 1024 > $GRAFT_REPO_ROOT/$O/prof$ov.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 $O/prof$ov.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('overlap $ov profiled ms/step', round(d['ms_per_step'],4))"
DB=$(find $O/prof$ov -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats_overlap$ov.csv --steps 2044 | tail -50 > $O/stats$ov.txt
rm -rf $O/prof$ov
done
