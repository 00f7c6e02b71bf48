#!/bin/bash
# round 6 session 31: the experts' elementwise BatchNorm backward in the epilogue of the input-gradient contraction above
# (er_gemm_problem.bn_dz_out, HipBackend.frozen_dz_epilogue): tests, same-box A/B on MMoE
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s31; mkdir -p $O
timeout 1800 python -m pytest tests/test_fused_epilogues_gpu.py tests/test_models_gpu.py tests/test_kernels_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -12 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:18]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2; do
echo "mmoe25m_apart_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_FROZEN_DZ_EPILOGUE=0 line mmoe25m_apart_$rep --config configs/mmoe_taobao_4task_d64_25m.config $G
echo "mmoe25m_dz_epilogue_$rep" | tee -a $O/lines_summary.txt; line mmoe25m_dz_epilogue_$rep --config configs/mmoe_taobao_4task_d64_25m.config $G
done
echo mmoe25m_dz_epilogue_parity | tee -a $O/lines_summary.txt; line mmoe25m_dz_epilogue_parity --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 128 --cpu_seconds 2
ls $O | head -3
