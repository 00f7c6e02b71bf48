#!/bin/bash
# round 6 session 12: BatchNorm backward's partials 16 per trip, FM / rowsum loads batched: kernel tests + same-box A/B;
# the default step's kernel timeline (rocprofv3 --kernel-trace, rocpd) - durations by grid and the gaps between kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s12; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -6 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "default_prev_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line default_prev_$rep $F
echo "default_new_$rep" | tee -a $O/lines_summary.txt; line default_new_$rep $F
done
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo mmoe25m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line mmoe25m_prev --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_new | tee -a $O/lines_summary.txt; line mmoe25m_new --config configs/mmoe_taobao_4task_d64_25m.config $G
echo din10m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line din10m_prev --config configs/din_taobao_10m.config $G
echo din10m_new | tee -a $O/lines_summary.txt; line din10m_new --config configs/din_taobao_10m.config $G
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o trace -- python bench.py --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 64 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1); echo "db $DB"
python tools/rocpd_timeline.py $DB 100 > $O/default_step_timeline.txt 2>&1; tail -40 $O/default_step_timeline.txt
python tools/rocpd_by_grid.py $DB er:: > $O/default_kernels_by_grid.txt 2>&1
rm -rf $O/prof
ls $O; du -sh $O
