#!/bin/bash
# round 6 session 9: loads taken out of per-element branches (the compiler waited for each inside its branch):
# BatchNorm forward finalize (partials in groups of 16, pooled merge), the fp32 GEMM's generic (unaligned) tile fetch and its
# C += epilogue.  Kernel tests, then same-box A/B lines against the round's previous library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s9; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py tests/test_deepfm_gpu.py tests/test_models_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -15 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'), '| from_file', d.get('from_file'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "default_prev_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line default_prev_$rep $F
echo "default_new_$rep" | tee -a $O/lines_summary.txt; line default_new_$rep $F
done
echo default_new_parity_csv | tee -a $O/lines_summary.txt; line default_new_parity --steady_steps 0 --precondition 256 --cpu_seconds 2 --from_file csv
echo default_new_criteo | tee -a $O/lines_summary.txt; line default_new_criteo $F --from_file criteo
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo mmoe25m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line mmoe25m_prev --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_new | tee -a $O/lines_summary.txt; line mmoe25m_new --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 0 --precondition 128 --cpu_seconds 2
echo mmoe25m_new_mid4 | tee -a $O/lines_summary.txt; ER_BN_TILES_MID=4 line mmoe25m_new_mid4 --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_new_mid8 | tee -a $O/lines_summary.txt; ER_BN_TILES_MID=8 line mmoe25m_new_mid8 --config configs/mmoe_taobao_4task_d64_25m.config $G
echo din10m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line din10m_prev --config configs/din_taobao_10m.config $G
echo din10m_new | tee -a $O/lines_summary.txt; line din10m_new --config configs/din_taobao_10m.config --steady_steps 0 --precondition 128 --cpu_seconds 2
echo dcnv2_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line dcnv2_prev --config configs/dcn_v2_criteo.config $F
echo dcnv2_new | tee -a $O/lines_summary.txt; line dcnv2_new --config configs/dcn_v2_criteo.config --steady_steps 0 --precondition 256 --cpu_seconds 2
ls $O; du -sh $O
