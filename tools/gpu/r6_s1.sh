#!/bin/bash
# round 6 session 1: the fused cross layer + bf16 epilogues (tests/test_fused_epilogues_gpu.py), the DCN-v2 model tests, and
# same-box A/B lines of BASELINE config 3 (fp32 / bf16) with the fusions on and off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s1; mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1; nproc >> $O/device.txt
timeout 900 python -m pytest tests/test_fused_epilogues_gpu.py -q --timeout 300 -x 2>&1 | tail -15 | tee $O/tests_new.txt
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_kernels_gpu.py -q --timeout 300 -k "dcn or bf16 or cross or bn or concat" 2>&1 | tail -8 | tee $O/tests_models.txt
line() { name=$1; shift; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
Q="--config configs/dcn_v2_criteo.config --steady_steps 0 --precondition 256 --cpu_seconds 2"
echo dcnv2_f32_fused | tee -a $O/lines_summary.txt; line dcnv2_f32_fused $Q
echo dcnv2_f32_unfused | tee -a $O/lines_summary.txt; EASYREC_AMD_FUSED_CROSS=0 line dcnv2_f32_unfused $Q --no_cpu_baseline
echo dcnv2_bf16_fused | tee -a $O/lines_summary.txt; line dcnv2_bf16_fused $Q --dense_dtype bf16
echo dcnv2_bf16_unfused | tee -a $O/lines_summary.txt; EASYREC_AMD_FUSED_CROSS=0 EASYREC_AMD_BF16_EPILOGUES=0 line dcnv2_bf16_unfused $Q --dense_dtype bf16 --no_cpu_baseline
echo default | tee -a $O/lines_summary.txt; line default --steady_steps 0 --no_cpu_baseline --precondition 256
ls $O; du -sh $O
