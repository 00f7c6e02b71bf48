#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite, the default bench line, a kernel-trace profile of the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
rocminfo | grep -m2 -E "gfx|Compute Unit" > $O/device.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3
tail -3 $O/pytest.log; tail -c 1500 $O/bench.log
