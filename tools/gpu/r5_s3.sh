#!/bin/bash
# round 5 session 3: the dim-1 follower's tiles riding on the leader's in er_emb_bwd_fused (own_pair_tile_body) + the
# scratch-copy fix of the fix launch: embedding / DeepFM / model GPU tests, same-box A/B, per-kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_files_to_gpu.py tests/test_models_gpu.py tests/test_kv_embedding.py -q -m gpu -x --timeout 300 2>&1 | tail -6 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', (r.get('embedding_stage') or {}).get('us_per_step'), (r.get('embedding_stage') or {}).get('frac_of_hbm_peak'))
print('   ', ' | '.join('%s %.1f' % (k['kernel'][:28], k['us_per_step']) for k in r.get('kernels', []) if 'emb' in k['kernel']))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
EASYREC_AMD_PAIR_TILES=0 run own_tiles $Q
run paired_tiles $Q
EASYREC_AMD_PAIR_TILES=0 run own_tiles_again $Q
run paired_tiles_again $Q
EASYREC_AMD_DEFER_CATCH_UP=0 run paired_catch_up_launch $Q
run paired_uniform $Q --ids uniform
