#!/bin/bash
# round 5 session 6: the own launch's compile-time knobs as same-box A/B builds (tools/ab_variants.sh): entries per dim-1
# tile (256 / 512 / 1024), column-reduction parts per one-row table (16 / 8), waves per SIMD (4 / 5); + the fused-step tests
# on the cleaned-up library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s6; mkdir -p $O
timeout 600 python -m pytest tests/test_deepfm_gpu.py -q -m gpu -x --timeout 300 -k "fused or variants or replay_table or first_step or graph" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_diff', p), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4), '|', ' | '.join('%s %.1f' % (k['kernel'][:24], k['us_per_step']) for k in r.get('kernels', []) if 'emb' in k['kernel']))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
run default $Q
for v in t4 t4p8 t2p8 p8 w5; do EASYREC_AMD_LIB=$PWD/_ab/$v/libeasyrec_hip.so run $v $Q; done
run default_again $Q
for v in t4p8 t2p8; do EASYREC_AMD_LIB=$PWD/_ab/$v/libeasyrec_hip.so run ${v}_again $Q; done
