#!/bin/bash
# split-K granularity of the weight-gradient launch after the XCD rotation: rows per split / split cap / block target,
# default kernel and natural-layout kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "natural or grouped" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step |', ' | '.join('%s %.1f' % (k['kernel'][14:50], k['us_per_step']) for k in (r.get('kernels') or [])[:40] if 'grouped' in k['kernel'] and ('false, false' in k['kernel'] or 'tnn' in k['kernel'] or 'reduce' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
D="--config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
F="--no_cpu_baseline --steady_steps 128 --parity_steps 0"
export ER_WGRAD_MAX_SPLITS=1024
for rows in 2048 1024 512 256; do
  for mode in 0 1; do
    ER_GEMM_TNN=$mode ER_WGRAD_SPLIT_ROWS=$rows run din_tnn${mode}_rows$rows $D
  done
done
for tb in 512 1024 2048; do
  ER_GEMM_TNN=0 ER_WGRAD_TARGET_BLOCKS=$tb run mmoe_tb$tb $M
  ER_GEMM_TNN=0 ER_WGRAD_TARGET_BLOCKS=$tb run deepfm_tb$tb $F
done
ER_GEMM_TNN=0 ER_WGRAD_SPLIT_ROWS=512 run mmoe_rows512 $M
ER_GEMM_TNN=0 ER_WGRAD_SPLIT_ROWS=512 run deepfm_rows512 $F
