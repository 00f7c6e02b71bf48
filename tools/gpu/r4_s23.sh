#!/bin/bash
# round 4 session 23: the sort's lane exchanges in the VALU (DPP / permlane swaps instead of ds_bpermute) and the concat's
# 16-byte row pitch: sort / routing / DeepFM tests, then same-box A/B (base library = the commit before: _ab/base)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s23; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py tests/test_embedding_parallel_gpu.py -q -m gpu -x --timeout 600 -k "sort or segment or route or routing or fused or deepfm or embedding_step or bwd" 2>&1 | tail -6 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in r.get('kernels', []):
  if any(t in k['kernel'] for t in ('front_sort', 'gemm_f32_kernel')): print('    ', k['kernel'][:60], k['launches_per_step'], round(k['us_per_step'],1))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 300 --warmup 20"
BASE=$GRAFT_REPO_ROOT/_ab/base/easyrec_amd/csrc/libeasyrec_hip.so
EASYREC_AMD_LIB=$BASE EASYREC_AMD_CONCAT_PITCH=0 run base $Q
EASYREC_AMD_CONCAT_PITCH=0 run valu_sort $Q
run valu_sort_pitch $Q
EASYREC_AMD_LIB=$BASE EASYREC_AMD_CONCAT_PITCH=0 run base_again $Q
run valu_sort_pitch_parity --no_cpu_baseline --steady_steps 0 --steps 100 --warmup 20
