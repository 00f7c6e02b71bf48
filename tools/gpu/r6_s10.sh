#!/bin/bash
# round 6 session 10: the embedding row update's gather with every load outside a branch (own_finish_issue / _combine: tile
# passes two at a time, one-row tables four rows per trip), lazy_row's record in one request, fix_body's keys in one request.
# Embedding tests, then same-box A/B against the library of the previous commit's embedding kernels.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s10; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_embedding_stage_pins.py tests/test_embedding_parallel_gpu.py tests/test_kv_embedding.py tests/test_files_to_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -12 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'), '| from_file', json.dumps(d.get('from_file'))[:600])
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "default_prev_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line default_prev_$rep $F
echo "default_new_$rep" | tee -a $O/lines_summary.txt; line default_new_$rep $F
done
echo default_new_parity_csv | tee -a $O/lines_summary.txt; line default_new_parity --steady_steps 0 --precondition 256 --cpu_seconds 2 --from_file csv
echo default_new_criteo | tee -a $O/lines_summary.txt; line default_new_criteo $F --from_file criteo
echo uniform_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line uniform_prev --ids uniform $F
echo uniform_new | tee -a $O/lines_summary.txt; line uniform_new --ids uniform $F
echo ep1_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line ep1_prev --force_ep --rccl $F
echo ep1_new | tee -a $O/lines_summary.txt; line ep1_new --force_ep --rccl $F
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo mmoe25m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line mmoe25m_prev --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_new | tee -a $O/lines_summary.txt; line mmoe25m_new --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 0 --precondition 128 --cpu_seconds 2
echo din10m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line din10m_prev --config configs/din_taobao_10m.config $G
echo din10m_new | tee -a $O/lines_summary.txt; line din10m_new --config configs/din_taobao_10m.config --steady_steps 0 --precondition 128 --cpu_seconds 2
ls $O; du -sh $O
