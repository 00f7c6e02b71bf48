#!/bin/bash
# round 4, session 4: cold instruction fetch probe; fused head third cut
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s4; mkdir -p $O
timeout 300 tools/micro/launch_floor 48 200 2>&1 | tee $O/launch_floor.txt | tail -6
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "head_sigmoid" --timeout 300 2>&1 | tail -5 | tee $O/tests_head.log
