#!/bin/bash
# are the split-K blocks of a batch-long contraction marching over HBM channels in lock step?  rows per split that are
# not a power of two (the distance between two blocks' streams then is no multiple of the channel interleave)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
export ER_WGRAD_MAX_SPLITS=1024
for rows in 2048 2080 2016 1056 544 4128 800; do echo "rows/split $rows" | tee -a $O/probe.log; for i in 0 2; do ER_WGRAD_SPLIT_ROWS=$rows timeout 300 python tools/wgrad_probe.py --only $i 2>&1 | grep TN | tee -a $O/probe.log; done; done
