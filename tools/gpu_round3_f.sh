#!/bin/bash
# round 3: the bench lines of BASELINE configs 3-5 and the default line after the grouped stacks landed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_parallel_gpu.py -m gpu -q -x -k "closed_form" 2>&1 | tail -30 | tee $O/serve_test.txt
