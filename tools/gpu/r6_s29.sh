#!/bin/bash
# round 6 session 29: the weight gradient of DIN's [B x L, 32] -> [B x L, 1] score projection out of the grouped launch
# (er_wgrad_tall_narrow): kernel tests, DIN tests, same-box A/B on DIN (EASYREC_AMD_TALL_GEMV=0: tiles + grouped launch)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s29; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py tests/test_models_gpu.py -q --timeout 600 -m gpu -k "tall or din or DIN or narrow" 2>&1 | tail -8 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:24]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2; do
echo "din10m_tiles_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_TALL_GEMV=0 line din10m_tiles_$rep --config configs/din_taobao_10m.config $G
echo "din10m_narrow_$rep" | tee -a $O/lines_summary.txt; line din10m_narrow_$rep --config configs/din_taobao_10m.config $G
done
echo din10m_narrow_parity | tee -a $O/lines_summary.txt; line din10m_narrow_parity --config configs/din_taobao_10m.config --steady_steps 64 --precondition 128 --cpu_seconds 2
ls $O
