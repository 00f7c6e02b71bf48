#!/bin/bash
# round 4, session 17: grad slots (no autograd add kernels in the cross stack), BatchNorm prefetch, full -m gpu suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s17; mkdir -p $O
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; s=d.get('steady_state') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20"
run deepfm $Q
EASYREC_AMD_GRAD_SLOTS=0 run dcnv2_f32_slots0 --config configs/dcn_v2_criteo.config $Q
run dcnv2_f32 --config configs/dcn_v2_criteo.config $Q
EASYREC_AMD_GRAD_SLOTS=0 run dcnv2_bf16_slots0 --config configs/dcn_v2_criteo.config --dense_dtype bf16 $Q
run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 $Q
run din --config configs/din_taobao_10m.config --precondition 128 $Q
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -6 | tee $O/pytest_gpu.log
