#!/bin/bash
# round 4, session 2: the fused binary head (er_head_sigmoid_ce + er_loss_tail): kernel test, model tests, same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "head_sigmoid" --timeout 300 2>&1 | tail -15 | tee $O/tests_head.log
timeout 1200 python -m pytest tests/test_deepfm_gpu.py tests/test_models_gpu.py tests/test_files_to_gpu.py -q -m gpu -x --timeout 600 2>&1 | tail -15 | tee $O/tests_models.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; s=d.get('steady_state') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | steady', round(s.get('ms_per_step_mean',0),4), '| parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20"
EASYREC_AMD_FUSED_HEAD=0 run deepfm_head0 $Q
run deepfm_head1 $Q
EASYREC_AMD_FUSED_HEAD=0 run deepfm_head0_again $Q
run deepfm_head1_parity --cpu_seconds 3 --steady_steps 0 --steps 200 --warmup 20
EASYREC_AMD_FUSED_HEAD=0 run dcnv2_head0 --config configs/dcn_v2_criteo.config $Q
run dcnv2_head1 --config configs/dcn_v2_criteo.config $Q
run dcnv2_bf16_head1 --config configs/dcn_v2_criteo.config --dense_dtype bf16 $Q
EASYREC_AMD_FUSED_HEAD=0 run din_head0 --config configs/din_taobao_10m.config --precondition 128 $Q
run din_head1 --config configs/din_taobao_10m.config --precondition 128 $Q
