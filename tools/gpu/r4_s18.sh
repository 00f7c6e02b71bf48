#!/bin/bash
# round 4, session 18: which aten kernels the DCN-v2 step still launches (tools/trace_step_ops.py), before the gradient slots
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s18; mkdir -p $O
timeout 300 python tools/trace_step_ops.py configs/dcn_v2_criteo.config 4096 2>&1 | grep -v "^lib" | tee $O/dcnv2_aten_ops.txt | tail -30
