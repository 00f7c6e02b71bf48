#!/bin/bash
# round 5 session 14: the deep tower's BatchNorm-backward sums from the final DNN's dgrad epilogue (er_gemm_f32_bn_bwd_cols,
# EASYREC_AMD_BN_COLS_EPILOGUE) and the embedding backward's narrow groups first (EASYREC_AMD_OWN_LONG_FIRST): tests, A/B lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s14; mkdir -p $O
timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_kernels_gpu.py -q -m gpu --timeout 300 -k "test_deepfm_gpu or column_block or batchnorm_backward_sums or gemm_grouped" 2>&1 | tail -8 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:30], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if ('emb' in k['kernel'] or 'bn_bwd' in k['kernel'] or 'gemm_f32_kernel<true, true>' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
run default $Q --parity_steps 2
EASYREC_AMD_BN_COLS_EPILOGUE=0 run no_cols $Q
EASYREC_AMD_OWN_LONG_FIRST=1 run long_first $Q
run default_again $Q
EASYREC_AMD_BN_COLS_EPILOGUE=0 run no_cols_again $Q
EASYREC_AMD_OWN_LONG_FIRST=1 run long_first_again $Q
