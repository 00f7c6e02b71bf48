#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/models; mkdir -p $O
timeout 1500 python -m pytest tests/test_models_gpu.py -m gpu -q --timeout 900 -s > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; grep -E "ms/step|passed|failed|Error|assert" $O/pytest.log | head -40; tail -5 $O/pytest.log
