#!/bin/bash
# (1) does a sustained run of one GEMM slow down against a burst of 30 launches (clocks)?  (2) DCN-v2 bf16 against fp32 now
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
for it in 30 300 3000 10000; do timeout 120 tools/micro/lib_gemm 0 8192 256 1152 0 $it 2>&1 | tee -a $O/sustained.log; done
for it in 30 3000; do timeout 120 tools/micro/gemm_core 1 8192 256 1152 $it 2>&1 | tee -a $O/sustained.log; done
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', d.get('dtype'), '| parity', (d.get('parity_full_size') or {}).get('max_rel_loss_diff'))
for f in (r.get('families') or []): print('   ', f.get('family'), round(f.get('us_per_step'),1), f.get('launches_per_step'))
for k in (r.get('kernels') or [])[:12]: print('      ', k['kernel'][:70], k['launches_per_step'], round(k['us_per_step'],1))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
C="--config configs/dcn_v2_criteo.config --steady_steps 128 --precondition 128 --no_cpu_baseline"
run dcnv2_bf16 $C --dense_dtype bf16
run dcnv2_f32 $C --parity_steps 0
