#!/bin/bash
# round 4, session 15: per-workgroup timestamp probe of emb_bwd_own_kernel (tools/own_probe.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s15; mkdir -p $O
timeout 300 python tools/own_probe.py 2>&1 | tail -20 | tee $O/probe.txt
