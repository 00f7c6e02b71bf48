#!/bin/bash
# round 5 session 12: the tail's riders (loss tail in the first grid, dense optimizer + split-K reduce behind the fix:
# er_emb_bwd_fused_tail): bit-identity tests, then same-box lines with EASYREC_AMD_TAIL_RIDERS on / off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s12; mkdir -p $O
timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_kernels_gpu.py tests/test_files_to_gpu.py -q -m gpu --timeout 300 -x -k "test_deepfm_gpu or gemm_grouped or loss_tail or head or dense_opt or test_files_to_gpu" 2>&1 | tail -8 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:30], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if ('emb' in k['kernel'] or 'grouped' in k['kernel'] or 'dense_opt' in k['kernel'] or 'loss_tail' in k['kernel'])))
print('    unfused', json.dumps(r.get('tail_unfused')), r.get('roofline_error'), d.get('roofline_error'))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
run riders $Q --parity_steps 2
EASYREC_AMD_TAIL_RIDERS=0 run no_riders $Q
run riders_again $Q
EASYREC_AMD_TAIL_RIDERS=0 run no_riders_again $Q
EASYREC_AMD_FUSED_TAIL=0 run no_tail $Q
run dcnv2 $Q --config configs/dcn_v2_criteo.config
EASYREC_AMD_TAIL_RIDERS=0 run dcnv2_no_riders $Q --config configs/dcn_v2_criteo.config
