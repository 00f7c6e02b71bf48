#!/bin/bash
# round 6 session 4: the embedding-parallel requester's merged tail (er_emb_reduce_local_tail): W = 1 - 4 GPU tests, A/B
# lines of `--force_ep --rccl`; DIN's generated-operand weight gradient by workgroup target
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s4; mkdir -p $O
timeout 1200 python -m pytest tests/test_embedding_parallel_gpu.py tests/test_multi_rank_oracle_gpu.py -q --timeout 600 -m gpu -x 2>&1 | tail -8 | tee $O/tests_ep.txt
timeout 600 python -m pytest tests/test_kv_embedding.py tests/test_fused_epilogues_gpu.py -q --timeout 300 -m gpu -k "hash_table_sequence or din" 2>&1 | tail -4 | tee $O/tests_misc.txt
line() { name=$1; shift; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:12]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
E="--force_ep --rccl --no_cpu_baseline --steady_steps 0 --precondition 256"
echo ep1_rccl_merged | tee -a $O/lines_summary.txt; line ep1_rccl_merged $E
echo ep1_rccl_apart | tee -a $O/lines_summary.txt; EASYREC_AMD_EP_MERGED_REDUCE=0 line ep1_rccl_apart $E
echo ep1_local_merged | tee -a $O/lines_summary.txt; line ep1_local_merged --force_ep --no_cpu_baseline --steady_steps 0 --precondition 256
Q="--config configs/din_taobao_10m.config --steady_steps 0 --precondition 128 --no_cpu_baseline"
for nb in 1024 512 2048; do echo din10m_wgrad_blocks_$nb | tee -a $O/lines_summary.txt; ER_DIN_WGRAD_BLOCKS=$nb line din_$nb $Q; done
ls $O; du -sh $O
