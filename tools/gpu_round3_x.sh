#!/bin/bash
# ReLU mask recomputed from z in the BatchNorm backward (EASYREC_AMD_BN_RECOMPUTE_MASK) A/B + tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "bn or batchnorm or relu_mask or linear" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step | bn', ' '.join('%.1f' % f['us_per_step'] for f in r.get('families', []) if f['family']=='batchnorm'), '| gemm', ' '.join('%.1f' % f['us_per_step'] for f in r.get('families', []) if f['family']=='gemm'))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
D="--config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
F="--no_cpu_baseline --steady_steps 128 --parity_steps 0"
for v in 1 0; do
  EASYREC_AMD_BN_RECOMPUTE_MASK=$v run din_mask$v $D
  EASYREC_AMD_BN_RECOMPUTE_MASK=$v run mmoe_mask$v $M
  EASYREC_AMD_BN_RECOMPUTE_MASK=$v run deepfm_mask$v $F
done
