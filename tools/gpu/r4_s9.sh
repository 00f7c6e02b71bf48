#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s9; mkdir -p $O
ER_HEAD_TS=1 timeout 60 tools/micro/lib_chain head 32 100 2>&1 | tee -a $O/lib_chain.txt
