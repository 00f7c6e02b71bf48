#!/bin/bash
# round 6 session 28: frozen BatchNorm backward in one pass (ER_BN_FROZEN_ONE_PASS): same-box A/B on MMoE; the GPU suites of the
# kernel / model / epilogue files on the tree; DeepFM and DCN-v2 lines (no regression)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s28; mkdir -p $O
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:18]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2; do
echo "mmoe25m_two_pass_$rep" | tee -a $O/lines_summary.txt; ER_BN_FROZEN_ONE_PASS=0 line mmoe25m_two_pass_$rep --config configs/mmoe_taobao_4task_d64_25m.config $G
echo "mmoe25m_one_pass_$rep" | tee -a $O/lines_summary.txt; line mmoe25m_one_pass_$rep --config configs/mmoe_taobao_4task_d64_25m.config $G
done
echo mmoe25m_parity | tee -a $O/lines_summary.txt; line mmoe25m_parity --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 128 --cpu_seconds 2
echo default | tee -a $O/lines_summary.txt; line default --no_cpu_baseline --steady_steps 0 --precondition 256
echo dcnv2_f32 | tee -a $O/lines_summary.txt; line dcnv2_f32 --config configs/dcn_v2_criteo.config $G
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py tests/test_models_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -8 | tee $O/tests.txt
ls $O
