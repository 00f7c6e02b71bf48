#!/bin/bash
# GPU session: full -m gpu suite, smoke, bench (with cpu_baseline), rocprofv3 kernel stats (CSV) and
# separate PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs, no trace domains besides kernel-trace).
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-s2}
O=gpurun_out/$TAG
mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1
nproc >> $O/device.txt; free -g | head -2 >> $O/device.txt; lscpu | grep "Model name" >> $O/device.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke exit $?" >> $O/smoke.log
tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench.log 2>&1
echo "bench exit $?" >> $O/bench.log
tail -3 $O/bench.log
timeout 600 python bench.py --optimizer lazy_adam --no_cpu_baseline > $O/bench_lazy.log 2>&1
tail -2 $O/bench_lazy.log
timeout 600 python bench.py --ids uniform --no_cpu_baseline > $O/bench_uniform.log 2>&1
tail -2 $O/bench_uniform.log
# kernel trace + stats of the same bench command
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline > $O/prof.log 2>&1
echo "prof exit $?" >> $O/prof.log
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_lazy -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline --optimizer lazy_adam > $O/prof_lazy.log 2>&1
# PMC passes (own runs)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -o bench -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/pmc_fetch.log 2>&1
echo "pmc fetch exit $?" >> $O/pmc_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write -o bench -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/pmc_write.log 2>&1
echo "pmc write exit $?" >> $O/pmc_write.log
# keep only the small files (kernel trace csv of pmc runs can be large)
find $O -name "*.csv" -size +20M -delete
ls -laR $O | head -60
