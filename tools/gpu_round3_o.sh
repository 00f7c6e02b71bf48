#!/bin/bash
# tools/micro/tn_stream: loads only / + LDS / + MFMA of the natural-layout TN kernel against the number of workgroups
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
for v in 0 6 7 4 1 2 3 5; do
  for nb in 100 256 512 1024 3200; do
    timeout 60 tools/micro/tn_stream $v $nb 2>&1 | tee -a $O/tn_stream.log
  done
done
