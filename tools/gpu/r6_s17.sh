#!/bin/bash
# round 6 session 17: the device-wide sort's chunks by an LDS radix sort of (key, entry) pairs, 8192 entries per chunk
# (emb_chunk_radix_kernel); routed sort epilogue with keys / owners computed once: embedding / model tests, same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s17; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_embedding_parallel_gpu.py tests/test_embedding_stage_pins.py tests/test_kv_embedding.py tests/test_multi_rank_oracle_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -8 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
G="--no_cpu_baseline --steady_steps 0 --precondition 128"
echo mmoe25m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line mmoe25m_prev --config configs/mmoe_taobao_4task_d64_25m.config $G
echo mmoe25m_new | tee -a $O/lines_summary.txt; line mmoe25m_new --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 0 --precondition 128 --cpu_seconds 2
echo din10m_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line din10m_prev --config configs/din_taobao_10m.config $G
echo din10m_new | tee -a $O/lines_summary.txt; line din10m_new --config configs/din_taobao_10m.config --steady_steps 0 --precondition 128 --cpu_seconds 2
echo ep1_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line ep1_prev --force_ep --rccl $F
echo ep1_new | tee -a $O/lines_summary.txt; line ep1_new --force_ep --rccl $F
ls $O; du -sh $O
