#!/bin/bash
# round 4, session 10: shader / memory clocks while the bench runs (is any of the step clock-limited?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s10; mkdir -p $O
rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -5 | tee $O/clocks.txt
rocm-smi --showperflevel 2>&1 | grep -i perf | tee -a $O/clocks.txt
timeout 300 tools/micro/launch_floor 48 200 2>&1 | tee $O/launch_floor.txt | grep -E "^clock"
(timeout 100 python bench.py --no_cpu_baseline --parity_steps 0 --steady_steps 4096 --steps 2000 --warmup 20 > $O/bench.out 2>&1 &) ; sleep 45; for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -iE "sclk" | head -2; sleep 2; done | tee -a $O/clocks.txt; wait
