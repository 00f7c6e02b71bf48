#!/bin/bash
# round 5 session 19: tools/dbg_kv_seq_gpu.py (per-step GPU-vs-oracle loss differences of DIN-small, hash-table vs dense
# sequence tables, seeds, B = 48 / 512)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s19; mkdir -p $O
timeout 600 python tools/dbg_kv_seq_gpu.py 2>&1 | tail -20 | tee $O/probe.log
