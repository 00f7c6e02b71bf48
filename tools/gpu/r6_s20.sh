#!/bin/bash
# round 6 session 20: the BatchNorm finalize + apply of the layer in front of the head inside the head launch
# (er_head_sigmoid_ce_bn, EASYREC_AMD_HEAD_BN_APPLY): tests, same-box A/B on DeepFM / DIN
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s20; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_models_gpu.py tests/test_files_to_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -10 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "default_apart_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_HEAD_BN_APPLY=0 line default_apart_$rep $F
echo "default_fused_$rep" | tee -a $O/lines_summary.txt; line default_fused_$rep $F
done
echo default_fused_parity | tee -a $O/lines_summary.txt; line default_fused_parity --steady_steps 512 --precondition 256 --cpu_seconds 2
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo din10m_apart | tee -a $O/lines_summary.txt; EASYREC_AMD_HEAD_BN_APPLY=0 line din10m_apart --config configs/din_taobao_10m.config $G
echo din10m | tee -a $O/lines_summary.txt; line din10m --config configs/din_taobao_10m.config $G
ls $O
