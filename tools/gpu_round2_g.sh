#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest "tests/test_kernels_gpu.py::test_lazy_dense_decay_equals_the_sweep" "tests/test_deepfm_gpu.py::test_lazy_decay_equals_sweep_model_level" "tests/test_deepfm_gpu.py::test_evaluate_does_not_disturb_training" "tests/test_embedding_parallel_gpu.py::test_lazy_decay_equals_sweep_through_two_sharded_ranks" tests/test_embedding_parallel_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -20
echo "--- fresh tables, ring 16" | tee -a $O/ab.log
bash tools/gpu_ab2.sh "--steps 300 --steady_steps 0 --precondition 0 --ring 16" 2 | tee -a $O/ab.log
echo "--- preconditioned 1024, ring 256, steady 1024" | tee -a $O/ab.log
bash tools/gpu_ab2.sh "--steps 200 --steady_steps 1024" 2 | tee -a $O/ab.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 512 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $O/prof/step_results.db $O/kernel_stats.csv --steps 732 | tail -3; rm -rf $O/prof
