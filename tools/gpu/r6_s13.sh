#!/bin/bash
# round 6 session 13: the fp32 contraction's eight-wave k-split pair (gemm_f32_ks2_kernel) for launches of <= 256 tiles:
# whole GPU suite with it on, same-box A/B by ER_GEMM_KSPLIT, parity lines, the step timeline again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s13; mkdir -p $O
timeout 2400 python -m pytest tests -q --timeout 900 -m gpu 2>&1 | tail -25 | tee $O/tests_full.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "default_ks0_$rep" | tee -a $O/lines_summary.txt; ER_GEMM_KSPLIT=0 line default_ks0_$rep $F
echo "default_ks1_$rep" | tee -a $O/lines_summary.txt; line default_ks1_$rep $F
done
echo default_ks1_parity | tee -a $O/lines_summary.txt; line default_ks1_parity --steady_steps 0 --precondition 256 --cpu_seconds 2
echo dcnv2_ks0 | tee -a $O/lines_summary.txt; ER_GEMM_KSPLIT=0 line dcnv2_ks0 --config configs/dcn_v2_criteo.config $F
echo dcnv2_ks1 | tee -a $O/lines_summary.txt; line dcnv2_ks1 --config configs/dcn_v2_criteo.config --steady_steps 0 --precondition 256 --cpu_seconds 2
echo ep1_ks0 | tee -a $O/lines_summary.txt; ER_GEMM_KSPLIT=0 line ep1_ks0 --force_ep --rccl $F
echo ep1_ks1 | tee -a $O/lines_summary.txt; line ep1_ks1 --force_ep --rccl $F
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo mmoe25m_ks0 | tee -a $O/lines_summary.txt; ER_GEMM_KSPLIT=0 line mmoe25m_ks0 --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_ks1 | tee -a $O/lines_summary.txt; line mmoe25m_ks1 --config configs/mmoe_taobao_4task_d64_25m.config $G
echo din10m_ks0 | tee -a $O/lines_summary.txt; ER_GEMM_KSPLIT=0 line din10m_ks0 --config configs/din_taobao_10m.config $G
echo din10m_ks1 | tee -a $O/lines_summary.txt; line din10m_ks1 --config configs/din_taobao_10m.config $G
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o trace -- python bench.py --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 64 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB 100 > $O/default_step_timeline.txt 2>&1; tail -36 $O/default_step_timeline.txt | cut -c1-150
rm -rf $O/prof
ls $O; du -sh $O
