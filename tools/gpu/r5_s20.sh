#!/bin/bash
# round 5 session 20: the hash-table sequence test alone, with its assertion text
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5s20
timeout 300 python -m pytest "tests/test_kv_embedding.py::test_hash_table_sequence_features_match_the_oracle_on_the_gpu" -q -m gpu --timeout 300 2>&1 | grep -E "^E  |Error|passed|failed|test_kv_embedding.py:[0-9]+" | head -20 | tee gpurun_out/r5s20/kv.log
