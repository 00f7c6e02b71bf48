#!/bin/bash
# round 6 session 22: embedding-parallel step, fewer launches: owner ids + build + merge + lag-1 table as one
# (er_emb_owner_ids_merge), owner fix + replicated apply + dense optimizer as one (er_emb_owner_update_tail): EP tests, A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s22; mkdir -p $O
timeout 1800 python -m pytest tests/test_embedding_parallel_gpu.py tests/test_kv_embedding.py -q --timeout 900 -m gpu -x 2>&1 | tail -15 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '|', ('one graph' if 'one hipGraph' in d.get('config', {}).get('workload', '') else 'segments/plain'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "ep1_rccl_apart_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_EP_OWNER_FUSED=0 EASYREC_AMD_EP_UPDATE_TAIL=0 line ep1_rccl_apart_$rep --force_ep --rccl $F
echo "ep1_rccl_owner_fused_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_EP_UPDATE_TAIL=0 line ep1_rccl_owner_fused_$rep --force_ep --rccl $F
echo "ep1_rccl_both_$rep" | tee -a $O/lines_summary.txt; line ep1_rccl_both_$rep --force_ep --rccl $F
done
echo "ep1_rccl_both_parity" | tee -a $O/lines_summary.txt; line ep1_rccl_both_parity --force_ep --rccl --steady_steps 0 --precondition 256 --cpu_seconds 2
ls $O
