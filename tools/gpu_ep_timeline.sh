#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ep_tl; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline --force_ep > $O/prof.log 2>&1
python tools/trace_timeline.py $O/prof/bench_kernel_trace.csv hyper_select 3 > $O/timeline.txt
rm -f $O/prof/*kernel_trace.csv
tail -1 $O/timeline.txt
