#!/bin/bash
# round 5 session 15: four k-tiles of loads in flight in the fp32 GEMM core (-DER_GEMM_PREFETCH=4, _ab/pf4) against two:
# the GEMM / model tests on the variant build, then same-box lines of configs 2-5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s15; mkdir -p $O
EASYREC_AMD_LIB=$PWD/_ab/pf4/libeasyrec_hip.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py -q -m gpu --timeout 300 -k "gemm or dgrad or dense or linear or mlp or batchnorm or test_deepfm_gpu" 2>&1 | tail -6 | tee $O/tests_pf4.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| dom', (r.get('kernel') or '')[:40], round(r.get('us_per_step', 0), 1), round(r.get('frac') or 0, 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:36], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if ('gemm' in k['kernel'] or 'wgrad' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
PF4="EASYREC_AMD_LIB=$PWD/_ab/pf4/libeasyrec_hip.so"
run deepfm_pf2 $Q
env $PF4 bash -c "$(declare -f run line); O=$O; run deepfm_pf4 $Q --parity_steps 2"
run deepfm_pf2_again $Q
env $PF4 bash -c "$(declare -f run line); O=$O; run deepfm_pf4_again $Q"
for c in dcn_v2_criteo din_taobao_10m mmoe_taobao_4task_d64_25m; do
  run ${c}_pf2 $Q --config configs/$c.config
  env $PF4 bash -c "$(declare -f run line); O=$O; run ${c}_pf4 $Q --config configs/$c.config"
done
