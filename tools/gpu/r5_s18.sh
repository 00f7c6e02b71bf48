#!/bin/bash
# round 5 session 18: the same test on the tree the session started from (0253e39, checked out under _old/)
cd $GRAFT_REPO_ROOT/_old; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s18; mkdir -p $O
T="tests/test_kv_embedding.py::test_hash_table_sequence_features_match_the_oracle_on_the_gpu"
timeout 300 python -m pytest "$T" -q -m gpu --timeout 300 2>&1 | grep -E "AssertionError|passed|failed" | head -4 | tee -a $O/old_tree.log
