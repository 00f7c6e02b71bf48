#!/bin/bash
# natural-layout TN kernel (ER_GEMM_TNN = 0 / 1 / 2) A/B + its bit-equality test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm or grouped or natural" 2>&1 | tail -8 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| parity', (d.get('parity_full_size') or {}).get('max_rel_loss_diff'))
for f in (r.get('families') or [])[:2]: print('   ', f.get('family'), round(f.get('us_per_step'),1), f.get('launches_per_step'))
for k in (r.get('kernels') or [])[:40]:
  if 'grouped' in k['kernel']: print('      ', k['kernel'][:70], k['launches_per_step'], round(k['us_per_step'],1))
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
D="--config configs/din_taobao_10m.config --steady_steps 128 --precondition 128 --no_cpu_baseline --parity_steps 0"
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --no_cpu_baseline --parity_steps 0"
F="--no_cpu_baseline --steady_steps 256 --parity_steps 0"
for mode in 0 1 2; do
  ER_GEMM_TNN=$mode run din_tnn$mode $D
done
for mode in 0 2; do
  ER_GEMM_TNN=$mode run mmoe_tnn$mode $M
  ER_GEMM_TNN=$mode run deepfm_tnn$mode $F
done
ER_GEMM_TNN=2 run dcnv2_tnn2 --config configs/dcn_v2_criteo.config --steady_steps 256 --precondition 256 --no_cpu_baseline --parity_steps 0
ER_GEMM_TNN=0 run dcnv2_tnn0 --config configs/dcn_v2_criteo.config --steady_steps 256 --precondition 256 --no_cpu_baseline --parity_steps 0
