#!/bin/bash
# round 6 session 21: the embedding-parallel step as ONE hipGraph with the collectives inside (EASYREC_AMD_EP_WHOLE_GRAPH):
# EP GPU tests, same-box A/B through the world-1 RCCL group and with local copies, its step timeline; and the kernel-boundary
# gap of the default step against DCN-v2's in ONE session (r6s14 showed 2.4 us where r6s12 showed 1.45)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s21; mkdir -p $O
timeout 1500 python -m pytest tests/test_embedding_parallel_gpu.py -q --timeout 900 -m gpu 2>&1 | tail -6 | tee $O/tests.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '|', ('one graph' if 'one hipGraph' in d.get('config', {}).get('workload', '') else 'segments/plain'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2; do
echo "ep1_rccl_segments_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_EP_WHOLE_GRAPH=0 line ep1_rccl_segments_$rep --force_ep --rccl $F
echo "ep1_rccl_whole_$rep" | tee -a $O/lines_summary.txt; line ep1_rccl_whole_$rep --force_ep --rccl $F
done
echo "ep1_rccl_whole_overlap" | tee -a $O/lines_summary.txt; EASYREC_AMD_EP_OVERLAP=1 line ep1_rccl_whole_overlap --force_ep --rccl $F
echo "ep1_local_segments" | tee -a $O/lines_summary.txt; EASYREC_AMD_EP_WHOLE_GRAPH=0 line ep1_local_segments --force_ep $F
echo "ep1_local_whole" | tee -a $O/lines_summary.txt; line ep1_local_whole --force_ep $F
echo "default" | tee -a $O/lines_summary.txt; line default $F
tl() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof_$name -o trace -- python bench.py --steps 100 --warmup 20 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 64 "$@" > $O/$name.log 2>&1
  DB=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocpd_timeline.py $DB 60 > $O/${name}_step_timeline.txt 2>&1; tail -2 $O/${name}_step_timeline.txt | cut -c1-200
  rm -rf $O/prof_$name; }
tl default
tl dcnv2_f32 --config configs/dcn_v2_criteo.config
tl ep1_rccl_whole --force_ep --rccl
ls $O; du -sh $O
