#!/bin/bash
# round 6 session 2: after (a) the slot gate stopped materialising zero gradients, (b) the cross epilogues' operands are
# requested before the k loop, (c) a bf16 step's weight gradients ride in the fp32 fused tail, (d) concat tags its BatchNorm
# block: tests, config-3 lines, and the FIRST per-kernel table of the embedding-parallel step since round 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s2; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_epilogues_gpu.py -q --timeout 300 -x 2>&1 | tail -5 | tee $O/tests_new.txt
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_deepfm_gpu.py -q --timeout 300 -k "dcn or bf16 or neighbouring or fused_step_variants or first_step" 2>&1 | tail -8 | tee $O/tests_models.txt
line() { name=$1; shift; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:16]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
Q="--config configs/dcn_v2_criteo.config --steady_steps 0 --precondition 256 --cpu_seconds 2"
echo dcnv2_f32 | tee -a $O/lines_summary.txt; line dcnv2_f32 $Q
echo dcnv2_bf16 | tee -a $O/lines_summary.txt; line dcnv2_bf16 $Q --dense_dtype bf16
echo dcnv2_bf16_wgrad_bf16 | tee -a $O/lines_summary.txt; EASYREC_AMD_BF16_WGRAD_F32=0 line dcnv2_bf16_wgrad_bf16 $Q --dense_dtype bf16 --no_cpu_baseline
echo ep1_rccl | tee -a $O/lines_summary.txt; line ep1_rccl --force_ep --rccl --no_cpu_baseline --steady_steps 0 --precondition 256
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o ep -- python bench.py --force_ep --rccl --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 128 > $O/prof_ep.log 2>&1
cp $O/prof/ep_kernel_stats.csv $O/kernel_stats_ep1_rccl.csv 2>/dev/null || cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_ep1_rccl.csv; rm -rf $O/prof
head -60 $O/kernel_stats_ep1_rccl.csv | cut -c1-200
ls $O; du -sh $O
