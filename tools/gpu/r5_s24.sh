#!/bin/bash
# round 5 session 24: smoke + the fused-step bit-identity test after the discard of stale deferred loss tails
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s24; mkdir -p $O
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120 | tee $O/smoke.txt
timeout 200 python -m pytest tests/test_deepfm_gpu.py -q -m gpu -k "fused_step_variants or deterministic" 2>&1 | tail -1 | tee $O/tests.txt
