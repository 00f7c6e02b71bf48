#!/bin/bash
# bf16 weight gradients in one grouped launch: DCN-v2 bf16 against fp32, + tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -m gpu -x -k "gemm or grouped or bf16" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', d.get('dtype'), '| parity', (d.get('parity_full_size') or {}).get('max_rel_loss_diff'))
for f in (r.get('families') or [])[:4]: print('   ', f.get('family'), round(f.get('us_per_step'),1), f.get('launches_per_step'))
for k in (r.get('kernels') or [])[:8]: print('      ', k['kernel'][:70], k['launches_per_step'], round(k['us_per_step'],1))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
C="--config configs/dcn_v2_criteo.config --steady_steps 128 --precondition 128 --no_cpu_baseline"
run dcnv2_bf16 $C --dense_dtype bf16
run dcnv2_f32 $C --parity_steps 0
