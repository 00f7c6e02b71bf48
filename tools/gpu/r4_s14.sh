#!/bin/bash
# round 4, session 14: fused embedding step, second cut: A/B test + bench A/B with the embedding kernels listed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s14; mkdir -p $O
timeout 900 python -m pytest tests/test_deepfm_gpu.py -q -m gpu -x -k "fused_embedding" --timeout 300 2>&1 | tail -3 | tee $O/tests_fused.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; s=d.get('steady_state') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | steady', round(s.get('ms_per_step_mean',0),4), '| parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in r.get('kernels', []):
  if any(t in k['kernel'] for t in ('emb_', 'hyper', 'hash', 'grad_finish', 'decay')): print('    ', k['kernel'][:60], k['us_per_step'])
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20"
run deepfm_fused $Q
