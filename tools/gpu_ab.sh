#!/bin/bash
# A/B on ONE box (box-to-box variance is ~5-10%): usage gpu_ab.sh "<env for A>" "<env for B>" [bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ab; mkdir -p $O
A="$1"; B="$2"; shift 2
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E timeout 600 python bench.py --no_cpu_baseline --steps 200 "$@" > $O/b.log 2>&1
    echo "$v [$E] $(tail -1 $O/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
  done
done
