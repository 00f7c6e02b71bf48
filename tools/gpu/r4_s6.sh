#!/bin/bash
# round 4, session 6: the launch-floor micro-benchmark (tools/micro/launch_floor.hip) after the ce-shape variants were added
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s6; mkdir -p $O
timeout 300 tools/micro/launch_floor 48 200 2>&1 | tee $O/launch_floor.txt | tail -12
