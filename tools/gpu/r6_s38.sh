#!/bin/bash
# round 6 session 38: BnBwdEpi.dz_out in its own instantiation (gemm_f32_grouped_bn_bwd_dz_kernel): inside the plain BN_EPI
# kernels the branch cost DeepFM's five input-gradient launches 40.8 -> 47.5 us.  Same-box A/B against the previous library
# (EASYREC_AMD_LIB) on DeepFM / DCN-v2 / DIN / MMoE, the epilogue tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s38; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py -q --timeout 600 -m gpu -k "gemm or frozen or bn or grouped or multi_task" 2>&1 | tail -4 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:50]:
  if 'bn_bwd' in k['kernel'] and 'gemm' in k['kernel']: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
for rep in 1 2; do
echo "default_before_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PWD/gpurun_prev/libeasyrec_hip_prev.so line default_before_$rep --no_cpu_baseline --steady_steps 0 --precondition 256
echo "default_after_$rep" | tee -a $O/lines_summary.txt; line default_after_$rep --no_cpu_baseline --steady_steps 0 --precondition 256
done
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo "dcnv2_before" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PWD/gpurun_prev/libeasyrec_hip_prev.so line dcnv2_before --config configs/dcn_v2_criteo.config $G
echo "dcnv2_after" | tee -a $O/lines_summary.txt; line dcnv2_after --config configs/dcn_v2_criteo.config $G
echo "din10m_before" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PWD/gpurun_prev/libeasyrec_hip_prev.so line din10m_before --config configs/din_taobao_10m.config $G
echo "din10m_after" | tee -a $O/lines_summary.txt; line din10m_after --config configs/din_taobao_10m.config $G
echo "mmoe25m_before" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PWD/gpurun_prev/libeasyrec_hip_prev.so line mmoe25m_before --config configs/mmoe_taobao_4task_d64_25m.config $G
echo "mmoe25m_after" | tee -a $O/lines_summary.txt; line mmoe25m_after --config configs/mmoe_taobao_4task_d64_25m.config $G
