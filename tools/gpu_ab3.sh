#!/bin/bash
# same-box comparison of several library builds: gpu_ab3.sh lib1 lib2 ... (paths relative to the repo)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ab; mkdir -p $O
for rep in 1 2; do
  for lib in "$@"; do
    EASYREC_AMD_LIB=$PWD/$lib timeout 600 python bench.py --no_cpu_baseline --steps 200 > $O/b.log 2>&1
    echo "$lib $(tail -1 $O/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
  done
done
