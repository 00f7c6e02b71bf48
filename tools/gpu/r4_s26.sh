#!/bin/bash
# round 4 session 26: the -m gpu suite on the final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s26; mkdir -p $O
timeout 380 python -m pytest tests -m gpu -q --timeout 300 -x --deselect "tests/test_models_gpu.py::test_full_size_parity_with_the_oracle" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
