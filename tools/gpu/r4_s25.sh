#!/bin/bash
# round 4 session 25: ONE weight-gradient GEMM for the readers of a shared input (MMoE's first depth): model tests, A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s25; mkdir -p $O
timeout 1200 python -m pytest tests/test_models_gpu.py -q -m gpu -x --timeout 600 -k "mmoe or multi_task or ple or dbmtl or full_size" 2>&1 | tail -4 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in r.get('kernels', [])[:6]: print('    ', k['kernel'][:70], k['launches_per_step'], round(k['us_per_step'],1))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 100 --warmup 10 --precondition 64"
EASYREC_AMD_CAT_WGRAD=0 run mmoe_grouped_wgrad --config configs/mmoe_taobao_4task_d64_25m.config $Q
run mmoe_cat_wgrad --config configs/mmoe_taobao_4task_d64_25m.config $Q
EASYREC_AMD_CAT_WGRAD=0 run mmoe_grouped_wgrad_again --config configs/mmoe_taobao_4task_d64_25m.config $Q
