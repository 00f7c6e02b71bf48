#!/bin/bash
# round 5 session 25: the CSV -> device tests with the threaded host decoder on the GPU box's host
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s25; mkdir -p $O
timeout 100 python -m pytest tests/test_files_to_gpu.py tests/test_input_formats.py -q --timeout 90 -k "csv" 2>&1 | tail -2 | tee $O/tests.txt
