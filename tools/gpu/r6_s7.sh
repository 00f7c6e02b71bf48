#!/bin/bash
# round 6 session 7: BatchNorm forward finalize with the division-free pooled merge (bn_finalize_apply_body) - kernel tests,
# then same-box A/B lines against the previous library (EASYREC_AMD_LIB) on DeepFM / MMoE 25 M / DIN 10 M, and the
# rows-per-workgroup knob (ER_BN_TILES_MID) for B = 8192 under the cheaper merge
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s7; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py tests/test_deepfm_gpu.py -q --timeout 600 -m gpu -x 2>&1 | tail -5 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
Q="--steady_steps 0 --precondition 128 --cpu_seconds 2"
for rep in 1 2; do
echo "default_prev_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line default_prev_$rep --no_cpu_baseline --steady_steps 0 --precondition 256
echo "default_new_$rep" | tee -a $O/lines_summary.txt; line default_new_$rep --steady_steps 0 --precondition 256 --cpu_seconds 2
done
echo mmoe25m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line mmoe25m_prev --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 0 --precondition 128
echo mmoe25m_new | tee -a $O/lines_summary.txt; line mmoe25m_new --config configs/mmoe_taobao_4task_d64_25m.config $Q
echo mmoe25m_new_mid1 | tee -a $O/lines_summary.txt; ER_BN_TILES_MID=1 line mmoe25m_new_mid1 --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 0 --precondition 128
echo mmoe25m_new_mid4 | tee -a $O/lines_summary.txt; ER_BN_TILES_MID=4 line mmoe25m_new_mid4 --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 0 --precondition 128
echo din10m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line din10m_prev --config configs/din_taobao_10m.config --no_cpu_baseline --steady_steps 0 --precondition 128
echo din10m_new | tee -a $O/lines_summary.txt; line din10m_new --config configs/din_taobao_10m.config $Q
ls $O; du -sh $O
