#!/bin/bash
# round 6 session 16: concat by 16-byte lanes (er_concat_cols, aligned blocks), BatchNorm-backward partial sums 8 rows per
# trip (bn_bwd_partial_body): kernel + model tests, same-box A/B against the previous commit's library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s16; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py tests/test_models_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -8 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2; do
echo mmoe25m_prev_$rep | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line mmoe25m_prev_$rep --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_new_$rep | tee -a $O/lines_summary.txt; line mmoe25m_new_$rep --config configs/mmoe_taobao_4task_d64_25m.config $G
done
echo din10m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line din10m_prev --config configs/din_taobao_10m.config $G
echo din10m_new | tee -a $O/lines_summary.txt; line din10m_new --config configs/din_taobao_10m.config $G
echo dcnv2_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line dcnv2_prev --config configs/dcn_v2_criteo.config $F
echo dcnv2_new | tee -a $O/lines_summary.txt; line dcnv2_new --config configs/dcn_v2_criteo.config $F
echo dcnv2_bf16_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line dcnv2_bf16_prev --config configs/dcn_v2_criteo.config --dense_dtype bf16 $F
echo dcnv2_bf16_new | tee -a $O/lines_summary.txt; line dcnv2_bf16_new --config configs/dcn_v2_criteo.config --dense_dtype bf16 $F
ls $O; du -sh $O
