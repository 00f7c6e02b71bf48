#!/bin/bash
# round 5 session 22: the whole -m gpu suite on the round's final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s22; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee $O/tests.log
