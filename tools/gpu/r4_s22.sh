#!/bin/bash
# round 4 session 22: DIN's first attention layer folded (K8b): kernel + model tests, same-box A/B on config 4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s22; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_din_paths.py -q -m gpu -x --timeout 600 -k "din or folded" 2>&1 | tail -12 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --steady_steps 0 --steps 100 --warmup 10 --precondition 64"
EASYREC_AMD_DIN_FOLD=0 run din_concat --config configs/din_taobao_10m.config $Q --parity_steps 0
run din_folded --config configs/din_taobao_10m.config $Q
EASYREC_AMD_DIN_FOLD=0 run din_concat_again --config configs/din_taobao_10m.config $Q --parity_steps 0
