#!/bin/bash
# round 5 session 13: test_fused_embedding_step_matches_the_general_path bisected over the session's switches (the tail's
# k-split count is what moved the step-3 loss), then the rest of the DeepFM GPU tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s13; mkdir -p $O
T="tests/test_deepfm_gpu.py::test_fused_embedding_step_matches_the_general_path"
for v in "" "EASYREC_AMD_TAIL_BLOCKS=0" "EASYREC_AMD_TAIL_RIDERS=0" "EASYREC_AMD_FUSED_TAIL=0" "ER_WGRAD_XCD=0"; do
  echo "=== variant [$v]" | tee -a $O/bisect.log
  env $v timeout 300 python -m pytest "$T" -q -m gpu --timeout 300 2>&1 | grep -E "AssertionError|assert |passed|failed|^E  " | head -12 | tee -a $O/bisect.log
done
timeout 600 python -m pytest tests/test_deepfm_gpu.py -q -m gpu --timeout 300 --deselect "$T" 2>&1 | tail -5 | tee -a $O/bisect.log
