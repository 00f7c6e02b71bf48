#!/bin/bash
# first GPU session: kernel parity tests, e2e tests, smoke, short bench
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt 2>&1
nproc >> gpurun_out/device.txt; free -g | head -2 >> gpurun_out/device.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
tail -30 gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_deepfm_gpu.py -m gpu -q --timeout 600 > gpurun_out/deepfm.log 2>&1
echo "deepfm exit $?" >> gpurun_out/deepfm.log
tail -30 gpurun_out/deepfm.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/bench1.log 2>&1
echo "bench exit $?" >> gpurun_out/bench1.log
tail -5 gpurun_out/bench1.log
# profile of the same bench command (kernel trace + stats)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/prof.log 2>&1
echo "prof exit $?" >> gpurun_out/prof.log
ls -R gpurun_out/prof | head -20
timeout 900 python bench.py --steps 50 --warmup 10 --optimizer lazy_adam --no_cpu_baseline > gpurun_out/bench_lazy.log 2>&1
tail -2 gpurun_out/bench_lazy.log
