#!/bin/bash
# round 6 session 35: the tall layers' BatchNorm partial merges (bn_stats_merge / bn_bwd_merge) four partials per trip:
# BatchNorm tests, same-box A/B on DIN against the previous library build (EASYREC_AMD_LIB)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s35; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogues_gpu.py -q --timeout 600 -m gpu -k "bn or batchnorm or staging or tall or din" 2>&1 | tail -4 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:40]:
  if 'merge_kernel' in k['kernel']: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
for rep in 1 2; do
echo "din10m_before_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PWD/gpurun_prev/libeasyrec_hip_prev.so line din10m_before_$rep --config configs/din_taobao_10m.config $G
echo "din10m_after_$rep" | tee -a $O/lines_summary.txt; line din10m_after_$rep --config configs/din_taobao_10m.config $G
done
echo din10m_parity | tee -a $O/lines_summary.txt; line din10m_parity --config configs/din_taobao_10m.config --steady_steps 64 --precondition 128 --cpu_seconds 2
