#!/bin/bash
# round 6 session 5: embedding-parallel GPU tests after the thread-local loss tail + the merged-tail bit-identity test;
# the --force_ep --rccl line with its family table; default line with clocks + --from_file csv / criteo
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s5; mkdir -p $O
timeout 1500 python -m pytest tests/test_embedding_parallel_gpu.py tests/test_multi_rank_oracle_gpu.py -q --timeout 600 -m gpu 2>&1 | tail -8 | tee $O/tests_ep.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'), '| from_file', d.get('from_file'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:30]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
echo ep1_rccl | tee -a $O/lines_summary.txt; line ep1_rccl --force_ep --rccl --no_cpu_baseline --steady_steps 0 --precondition 256
echo default_from_csv | tee -a $O/lines_summary.txt; line default_csv --steady_steps 0 --no_cpu_baseline --precondition 256 --from_file csv
echo default_from_criteo | tee -a $O/lines_summary.txt; line default_criteo --steady_steps 0 --no_cpu_baseline --precondition 256 --from_file criteo
ls $O; du -sh $O
