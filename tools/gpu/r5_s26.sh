#!/bin/bash
# round 5 session 26: the Criteo-binary / Parquet -> device tests after the native decimal packing of integer id columns
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s26; mkdir -p $O
timeout 60 python -m pytest tests/test_files_to_gpu.py -q --timeout 50 -k "criteo or parquet" 2>&1 | tail -2 | tee $O/tests.txt
