#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/knob; mkdir -p $O
python bench.py --no_cpu_baseline --no_overlap --steps 60 > $O/seq.log 2>&1
for k in 1 2 3 4 6 8; do
  ER_SWEEP_BLOCKS_PER_CU=$k python bench.py --no_cpu_baseline --steps 60 > $O/k$k.log 2>&1
done
for f in seq k1 k2 k3 k4 k6 k8; do echo -n "$f "; grep '^{' $O/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'])"; done
