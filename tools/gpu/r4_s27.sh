#!/bin/bash
# round 4 session 27: DeepFM's [sum(wide) | FM | deep] as one launch (er_wide_fm_concat): bit-exact kernel test, DeepFM model
# tests, same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s27; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_models_gpu.py -q -m gpu -x --timeout 150 -k "wide_fm or deepfm_matches or test_first or fused_embedding or graph or neighbouring_models_match_oracle and deepfm" 2>&1 | tail -4 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step |', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 200 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
EASYREC_AMD_FUSED_WIDE_FM=0 run three_launches $Q
run one_launch $Q
EASYREC_AMD_FUSED_WIDE_FM=0 run three_launches_again $Q
