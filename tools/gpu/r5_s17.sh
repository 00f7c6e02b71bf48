#!/bin/bash
# round 5 session 17: the hash-table sequence test's loss mismatch: which step, how far, and whether the weight gradients'
# split counts (ER_WGRAD_XCD) are what moved it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s17; mkdir -p $O
T="tests/test_kv_embedding.py::test_hash_table_sequence_features_match_the_oracle_on_the_gpu"
for v in "" "ER_WGRAD_XCD=0" "EASYREC_AMD_BN_COLS_EPILOGUE=0"; do
  echo "=== variant [$v]" | tee -a $O/bisect.log
  env $v timeout 300 python -m pytest "$T" -q -m gpu --timeout 300 2>&1 | grep -E "AssertionError|passed|failed" | head -4 | tee -a $O/bisect.log
done
