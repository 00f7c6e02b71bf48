#!/bin/bash
# full GPU suite, then a kernel trace of the embedding-parallel step (bench.py --force_ep) summarised by kernel shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ep_trace; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline --force_ep > $O/prof.log 2>&1
python tools/trace_summary.py $O/prof/bench_kernel_trace.csv > $O/all_by_shape.txt
rm -f $O/prof/*kernel_trace.csv
head -60 $O/all_by_shape.txt | cut -c1-200
