#!/bin/bash
# round 5 session 9: whole k-splits per XCD in the grouped weight-gradient launch (ER_WGRAD_XCD) and the step's tail in one
# grid (EASYREC_AMD_FUSED_TAIL): the tests that hold them bit-identical, then same-box A/B lines (DeepFM; DIN / MMoE default)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s9; mkdir -p $O
timeout 600 python -m pytest tests/test_deepfm_gpu.py tests/test_kernels_gpu.py -q -m gpu --timeout 300 -k "test_deepfm_gpu or gemm_grouped" 2>&1 | tail -8 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:30], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if ('emb' in k['kernel'] or 'grouped' in k['kernel'] or 'dense_opt' in k['kernel'])))
print('    tail', json.dumps(r.get('tail')), '| unfused', json.dumps(r.get('tail_unfused')), r.get('roofline_error'), d.get('roofline_error'))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
ER_WGRAD_XCD=0 EASYREC_AMD_FUSED_TAIL=0 run r4_tail $Q
EASYREC_AMD_FUSED_TAIL=0 run xcd_only $Q
run xcd_fused_tail $Q --parity_steps 2
ER_WGRAD_XCD=0 run fused_tail_only $Q
ER_WGRAD_TARGET_BLOCKS=1024 run xcd_fused_tail_1024 $Q
ER_WGRAD_TARGET_BLOCKS=256 run fused_tail_256 $Q
ER_WGRAD_XCD=0 EASYREC_AMD_FUSED_TAIL=0 run r4_tail_again $Q
run xcd_fused_tail_again $Q
run din10m $Q --config configs/din_taobao_10m.config
run mmoe25m $Q --config configs/mmoe_taobao_4task_d64_25m.config
