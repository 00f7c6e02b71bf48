#!/bin/bash
# embedding-parallel requester path: parity tests, then a same-box comparison of library builds on bench.py --force_ep
# usage: gpu_ep_ab.sh lib1 lib2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ep_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_parallel_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/tests.log
for rep in 1 2; do
  for lib in "$@"; do
    EASYREC_AMD_LIB=$PWD/$lib timeout 600 python bench.py --no_cpu_baseline --steps 200 --force_ep > $O/b.log 2>&1
    echo "$lib $(tail -1 $O/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
  done
done
