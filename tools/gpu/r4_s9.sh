#!/bin/bash
# round 4, session 9: the library chains (tools/micro/lib_chain.cpp) with in-kernel timestamps: where the head kernel spends its time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s9; mkdir -p $O
ER_HEAD_TS=1 timeout 60 tools/micro/lib_chain head 32 100 2>&1 | tee -a $O/lib_chain.txt
