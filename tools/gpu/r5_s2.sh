#!/bin/bash
# round 5 session 2: lazy dense decay caught up IN REGISTERS (er_emb_front(DEFER) + er_emb_fwd_lazy + inline catch-up in
# er_emb_bwd_fused: no catch-up launch, nothing written before the row update): the embedding / DeepFM / file / kv GPU tests,
# a same-box A/B against the catch-up launch, and the per-kernel table of the default line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_files_to_gpu.py tests/test_models_gpu.py tests/test_kv_embedding.py tests/test_embedding_stage_pins.py -q -m gpu -x --timeout 300 2>&1 | tail -6 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', (r.get('embedding_stage') or {}).get('us_per_step'), (r.get('embedding_stage') or {}).get('frac_of_hbm_peak'))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
EASYREC_AMD_DEFER_CATCH_UP=0 run catch_up_launch $Q
run in_registers $Q
EASYREC_AMD_DEFER_CATCH_UP=0 run catch_up_launch_again $Q
run in_registers_again $Q
run in_registers_uniform $Q --ids uniform
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 --parity_steps 0 > $O/prof.log 2>&1
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null || cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_default.csv; rm -rf $O/prof
head -30 $O/kernel_stats_default.csv | cut -c1-200
