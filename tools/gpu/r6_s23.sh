#!/bin/bash
# round 6 session 23: the panel form of short contractions (gemm_f32_panel_kernel, ER_GEMM_PANEL / ER_GEMM_PANEL_TILES):
# kernel + model tests, same-box A/B on DeepFM (also with the bound at 1024 tiles), DCN-v2, DIN, MMoE
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s23; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_models_gpu.py tests/test_fused_epilogues_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -12 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:12]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "default_loop_$rep" | tee -a $O/lines_summary.txt; ER_GEMM_PANEL=0 line default_loop_$rep $F
echo "default_panel_$rep" | tee -a $O/lines_summary.txt; line default_panel_$rep $F
done
echo "default_panel_1024tiles" | tee -a $O/lines_summary.txt; ER_GEMM_PANEL_TILES=1024 line default_panel_1024tiles $F
echo default_panel_parity | tee -a $O/lines_summary.txt; line default_panel_parity --steady_steps 256 --precondition 256 --cpu_seconds 2
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo dcnv2_f32_loop | tee -a $O/lines_summary.txt; ER_GEMM_PANEL=0 line dcnv2_f32_loop --config configs/dcn_v2_criteo.config $G
echo dcnv2_f32_panel | tee -a $O/lines_summary.txt; line dcnv2_f32_panel --config configs/dcn_v2_criteo.config $G
echo din10m_loop | tee -a $O/lines_summary.txt; ER_GEMM_PANEL=0 line din10m_loop --config configs/din_taobao_10m.config $G
echo din10m_panel | tee -a $O/lines_summary.txt; line din10m_panel --config configs/din_taobao_10m.config $G
echo mmoe25m_loop | tee -a $O/lines_summary.txt; ER_GEMM_PANEL=0 line mmoe25m_loop --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_panel | tee -a $O/lines_summary.txt; line mmoe25m_panel --config configs/mmoe_taobao_4task_d64_25m.config $G
ls $O
