#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -60 ) > $O/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -40
timeout 300 python tools/trace_step_ops.py > $O/step_ops.txt 2>&1; grep -c "^lib\|^aten" $O/step_ops.txt; grep "^aten" $O/step_ops.txt
timeout 600 python bench.py --no_cpu_baseline --steps 300 --steady_steps 0 --precondition 0 --ring 16 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ring16 fresh tables ms/step', round(d['ms_per_step'],4))"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 512 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 $O/prof.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('preconditioned ms/step', round(d['ms_per_step'],4))"
