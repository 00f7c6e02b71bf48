#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
timeout 1500 python -m pytest tests/test_deepfm_gpu.py tests/test_multi_rank_oracle_gpu.py tests/test_models_gpu.py tests/test_kernels_gpu.py "tests/test_embedding_parallel_gpu.py::test_lazy_decay_equals_sweep_through_two_sharded_ranks" -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-400 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -60
