#!/bin/bash
# round 6 session 37: the tall score projection's bias gradient from its weight gradient's pass (er_wgrad_tall_narrow(dbias)):
# kernel + DIN tests, DIN lines on the tree (the previous library has another ABI for this entry: before = the committed lines of session 36)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s37; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py tests/test_models_gpu.py -q --timeout 600 -m gpu -k "tall or narrow or din or DIN or full_size" 2>&1 | tail -4 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2 3; do
echo "din10m_$rep" | tee -a $O/lines_summary.txt; line din10m_$rep --config configs/din_taobao_10m.config $G
done
echo din10m_parity | tee -a $O/lines_summary.txt; line din10m_parity --config configs/din_taobao_10m.config --steady_steps 64 --precondition 128 --cpu_seconds 2
