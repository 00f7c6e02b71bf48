#!/bin/bash
# round 4 session 21: hash-table tables under embedding parallelism (bucket / unbucket kernels, sharded engine), the
# filtered translate kernel test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s21; mkdir -p $O
timeout 1200 python -m pytest tests/test_embedding_parallel_gpu.py tests/test_kv_embedding.py -q -m gpu --timeout 600 -k "kv" 2>&1 | tail -40 | tee $O/tests.log
