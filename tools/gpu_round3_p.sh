#!/bin/bash
# exact grids (no surplus workgroups): probe + benches, default kernel and natural-layout kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm or grouped or natural or linear or batchnorm or bn" 2>&1 | tail -4 | tee $O/tests.log
export ER_WGRAD_MAX_SPLITS=1024
for rows in 2048 512; do echo "rows/split $rows" | tee -a $O/probe.log; ER_WGRAD_SPLIT_ROWS=$rows timeout 300 python tools/wgrad_probe.py 2>&1 | grep TN | tee -a $O/probe.log; done
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step | gemm', ' '.join('%.1f' % f['us_per_step'] for f in r.get('families', []) if f['family']=='gemm'), '|', ' | '.join('%s %.1f' % (k['kernel'][14:50], k['us_per_step']) for k in (r.get('kernels') or [])[:40] if 'grouped' in k['kernel'] and ('false, false' in k['kernel'] or 'tnn' in k['kernel'] or 'reduce' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
D="--config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
F="--no_cpu_baseline --steady_steps 128 --parity_steps 0"
C="--config configs/dcn_v2_criteo.config --steady_steps 128 --precondition 128 --no_cpu_baseline --parity_steps 0"
for mode in 0 1 2; do
  for rows in 2048 512; do
    ER_GEMM_TNN=$mode ER_WGRAD_SPLIT_ROWS=$rows run din_tnn${mode}_rows$rows $D
  done
done
for mode in 0 2; do
  ER_GEMM_TNN=$mode run mmoe_tnn$mode $M
  ER_GEMM_TNN=$mode run deepfm_tnn$mode $F
  ER_GEMM_TNN=$mode run dcnv2_tnn$mode $C
done
