#!/bin/bash
# round 6 session 8: per-lookup LDS radix sort of the narrow composites (seg_radix_sort_narrow) - the sort / route / embedding
# tests (the sorted output must equal the bitonic network's bit for bit), then same-box A/B lines:
#   prev (bitonic sort, Chan merge) | sortonly (radix sort, Chan merge) | new (radix sort, pooled BatchNorm merge)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s8; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_prev.so
SORT=$GRAFT_REPO_ROOT/easyrec_amd/csrc/ab/libeasyrec_hip_sortonly.so
EASYREC_AMD_LIB=$SORT timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_embedding_parallel_gpu.py tests/test_embedding_stage_pins.py -q --timeout 600 -m gpu 2>&1 | tail -15 | tee $O/tests_sortonly.txt
EASYREC_AMD_LIB=$PREV timeout 600 python -m pytest tests/test_deepfm_gpu.py -q --timeout 600 -m gpu -k "fused_embedding_step_matches_the_general_path" 2>&1 | tail -15 | tee $O/tests_prev_general_path.txt
timeout 600 python -m pytest tests/test_deepfm_gpu.py -q --timeout 600 -m gpu -k "fused_embedding_step_matches_the_general_path" 2>&1 | tail -30 | tee $O/tests_new_general_path.txt
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| clocks', d.get('clocks'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:14]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
F="--no_cpu_baseline --steady_steps 0 --precondition 256"
for rep in 1 2 3; do
echo "default_prev_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line default_prev_$rep $F
echo "default_sortonly_$rep" | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$SORT line default_sortonly_$rep $F
echo "default_new_$rep" | tee -a $O/lines_summary.txt; line default_new_$rep $F
done
echo ep1_prev | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$PREV line ep1_prev --force_ep --rccl $F
echo ep1_sortonly | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$SORT line ep1_sortonly --force_ep --rccl $F
echo default_sortonly_parity | tee -a $O/lines_summary.txt; EASYREC_AMD_LIB=$SORT line default_sortonly_parity --steady_steps 0 --precondition 256 --cpu_seconds 2
ls $O; du -sh $O
