#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/gemm2; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -x -k "gemm or linear" > $O/pytest_gemm.log 2>&1; echo "exit $?" >> $O/pytest_gemm.log; tail -5 $O/pytest_gemm.log
python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; cat $O/gemm_bench.txt
timeout 900 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $O/prof.log 2>&1
python tools/trace_summary.py $O/prof/bench_kernel_trace.csv gemm > $O/gemm_by_shape.txt
rm -f $O/prof/*kernel_trace.csv
cat $O/gemm_by_shape.txt
timeout 600 python bench.py --no_cpu_baseline --steps 60 > $O/b.log 2>&1; tail -1 $O/b.log | cut -c1-400
