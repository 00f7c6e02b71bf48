#!/bin/bash
# round 6 session 30: k-split length of batch-long weight gradients (ER_WGRAD_SPLIT_ROWS / ER_WGRAD_MAX_SPLITS) on DIN and MMoE
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s30; mkdir -p $O
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:24]:
  if 'grouped_kernel<false' in k['kernel'] or 'splitk' in k['kernel'] or 'din_kernel<false' in k['kernel']: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 64 --steps 60"
for sr in 2048 1024 512; do for ms in 128 256 512; do
echo "din10m_rows${sr}_max${ms}" | tee -a $O/lines_summary.txt; ER_WGRAD_SPLIT_ROWS=$sr ER_WGRAD_MAX_SPLITS=$ms line din_${sr}_${ms} --config configs/din_taobao_10m.config $G
done; done
for sr in 2048 1024 512; do
echo "mmoe25m_rows${sr}_max256" | tee -a $O/lines_summary.txt; ER_WGRAD_SPLIT_ROWS=$sr ER_WGRAD_MAX_SPLITS=256 line mmoe_${sr} --config configs/mmoe_taobao_4task_d64_25m.config $G
done
ls $O | head -3
