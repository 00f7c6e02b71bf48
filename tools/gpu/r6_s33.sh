#!/bin/bash
# round 6 session 33: the bias of DIN's tall score projection in the contraction's epilogue (LinearFn with bias instead of a
# bias launch forward and three backward): DIN tests + same-box lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s33; mkdir -p $O
timeout 1200 python -m pytest tests/test_fused_epilogues_gpu.py tests/test_models_gpu.py -q --timeout 600 -m gpu -k "din or DIN or tall or full_size" 2>&1 | tail -6 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2; do
echo "din10m_tiles_$rep (EASYREC_AMD_TALL_GEMV=0: also the bias launch path)" | tee -a $O/lines_summary.txt; EASYREC_AMD_TALL_GEMV=0 line din10m_tiles_$rep --config configs/din_taobao_10m.config $G
echo "din10m_$rep" | tee -a $O/lines_summary.txt; line din10m_$rep --config configs/din_taobao_10m.config $G
done
echo din10m_parity | tee -a $O/lines_summary.txt; line din10m_parity --config configs/din_taobao_10m.config --steady_steps 64 --precondition 128 --cpu_seconds 2
