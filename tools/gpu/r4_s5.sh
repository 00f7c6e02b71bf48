#!/bin/bash
# round 4, session 5: what the head / tail / loss kernels cost by themselves (warm chain) and which part of the head pays
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s5; mkdir -p $O
for op in ce head head_nosrc tail; do timeout 60 tools/micro/lib_chain $op 32 100 2>&1 | tee -a $O/lib_chain.txt; done
for d in 1 2 4 3 7; do ER_HEAD_DEBUG=$d timeout 60 tools/micro/lib_chain head 32 100 2>&1 | tee -a $O/lib_chain.txt; done
