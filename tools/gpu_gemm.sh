#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/gemm; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -x -k "gemm or linear" > $O/pytest_gemm.log 2>&1; echo "exit $?" >> $O/pytest_gemm.log; tail -8 $O/pytest_gemm.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_all.log 2>&1; echo "exit $?" >> $O/pytest_all.log; tail -8 $O/pytest_all.log
timeout 600 python bench.py --no_cpu_baseline --steps 60 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-330
timeout 600 python bench.py --no_cpu_baseline --steps 60 --optimizer lazy_adam > $O/bench_lazy.log 2>&1; tail -1 $O/bench_lazy.log | cut -c1-330
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_lazy -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline --optimizer lazy_adam > $O/prof_lazy.log 2>&1
rm -f $O/prof_lazy/*kernel_trace.csv
