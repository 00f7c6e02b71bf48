#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02n; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "bf16" 2>&1 | grep -v "^$" | cut -c1-300 | tail -25 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -20
timeout 300 python tools/gemm_bf16_bench.py 2>&1 | tee $O/gemm_bf16.txt | tail -12
timeout 600 python -m pytest tests/test_models_gpu.py -m gpu -q --tb=short -k "bf16" 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 | tee $O/pytest_models.log | grep -E "^(FAILED|ERROR)|passed|failed|^E " | head
