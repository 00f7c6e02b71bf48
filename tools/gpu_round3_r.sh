#!/bin/bash
# tools/micro/gemm_core: tile / k-tile / scheduling variants of the f32 MFMA GEMM core; + the colsum / loss unrolls in MMoE
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
for shape in "8192 1152 256" "8192 256 1152" "204800 128 128" "4096 256 640" "8192 256 256"; do
  for v in 0 1 2 3 4 5 6 7; do
    timeout 60 tools/micro/gemm_core $v $shape 2>&1 | tee -a $O/gemm_core.log
  done
done
python tools/gemm_bench.py 2>&1 | tail -20 | tee $O/gemm_bench.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step |', ' | '.join('%s %.1f' % (k['kernel'][4:30], k['us_per_step']) for k in (r.get('kernels') or [])[:40] if 'colsum' in k['kernel'] or 'sigmoid' in k['kernel'] or 'bn_finalize' in k['kernel']))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
run mmoe --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50
run din --config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50
