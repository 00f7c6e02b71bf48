#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/trace; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $O/prof.log 2>&1
python tools/trace_summary.py $O/prof/bench_kernel_trace.csv gemm bn_ > $O/gemm_bn_by_shape.txt
python tools/trace_summary.py $O/prof/bench_kernel_trace.csv > $O/all_by_shape.txt
rm -f $O/prof/*kernel_trace.csv
python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1
cat $O/gemm_bn_by_shape.txt
