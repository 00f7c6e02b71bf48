#!/bin/bash
# round 4 session 20: the hash-table filters (filter_freq / steps_to_live), device gAUC, checkpoints after the KV load
# path change
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s20; mkdir -p $O
timeout 1200 python -m pytest tests/test_kv_embedding.py -q -m gpu --timeout 600 -k "filtered_translate or match_the_oracle_on_the_gpu" 2>&1 | tail -60 | tee $O/tests2.log
