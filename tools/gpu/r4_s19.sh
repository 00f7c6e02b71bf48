#!/bin/bash
# round 4, session 19: gradient slots + library concat: aten op list, DCN / backbone model tests, same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s19; mkdir -p $O
timeout 300 python tools/trace_step_ops.py configs/dcn_v2_criteo.config 4096 2>&1 | grep -v "^lib" | tee $O/dcnv2_aten_ops.txt | tail -8
timeout 900 python -m pytest tests/test_models_gpu.py -q -m gpu -x --timeout 600 -k "dcn or backbone or neighbouring" 2>&1 | tail -4 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20"
EASYREC_AMD_GRAD_SLOTS=0 run dcnv2_f32_slots0 --config configs/dcn_v2_criteo.config $Q
run dcnv2_f32 --config configs/dcn_v2_criteo.config $Q
run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 $Q
