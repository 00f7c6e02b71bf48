#!/bin/bash
# one-pass DIN pooling / concat-backward kernels: tests + DIN step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -m gpu -x -k "din" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', d.get('dtype'), '| parity', (d.get('parity_full_size') or {}).get('max_rel_loss_diff'))
for f in (r.get('families') or []): print('   ', f.get('family'), round(f.get('us_per_step'),1), f.get('launches_per_step'))
for k in (r.get('kernels') or [])[:40]:
  if 'din_' in k['kernel'] or 'rocprim' in k['kernel']: print('      ', k['kernel'][:70], k['launches_per_step'], round(k['us_per_step'],1))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
run din --config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --cpu_seconds 2 --steps 50
