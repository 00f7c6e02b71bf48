#!/bin/bash
# end-of-round evidence run: full -m gpu suite, smoke, bench lines, rocprofv3 kernel stats, PMC passes (own runs).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-final}; O=gpurun_out/$TAG; mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1; nproc >> $O/device.txt; lscpu | grep "Model name" >> $O/device.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log; tail -2 $O/bench.log | cut -c1-2000
timeout 600 python bench.py --dense_sweep --no_cpu_baseline > $O/bench_sweep.log 2>&1; tail -1 $O/bench_sweep.log | cut -c1-300
timeout 600 python bench.py --optimizer lazy_adam --no_cpu_baseline > $O/bench_lazy.log 2>&1; tail -1 $O/bench_lazy.log | cut -c1-300
timeout 600 python bench.py --ids uniform --no_cpu_baseline > $O/bench_uniform.log 2>&1; tail -1 $O/bench_uniform.log | cut -c1-300
timeout 600 python bench.py --force_ep --no_cpu_baseline > $O/bench_ep1.log 2>&1; tail -1 $O/bench_ep1.log | cut -c1-300
timeout 600 python bench.py --force_ep --rccl --no_cpu_baseline > $O/bench_ep1_rccl.log 2>&1; tail -1 $O/bench_ep1_rccl.log | cut -c1-400
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline > $O/prof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_sweep -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline --dense_sweep > $O/prof_sweep.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_ep1 -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline --force_ep > $O/prof_ep1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -o bench -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline --dense_sweep > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write -o bench -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline --dense_sweep > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/cal_fetch -o cal -- python tools/pmc_calibrate.py > $O/cal_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/cal_write -o cal -- python tools/pmc_calibrate.py > $O/cal_write.log 2>&1
rm -f $O/*/*kernel_trace.csv
python tools/summarize_pmc.py $O > $O/pmc_summary.json 2>&1; cat $O/pmc_summary.json | head -40
