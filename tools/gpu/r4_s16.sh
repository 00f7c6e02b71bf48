#!/bin/bash
# round 4, session 16: emb_bwd_own_kernel variants (run-following vs tile partials) inside the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s16; mkdir -p $O
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step')
for k in r.get('kernels', []):
  if any(t in k['kernel'] for t in ('emb_bwd_own',)): print('    ', k['kernel'][:60], k['us_per_step'])
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 100 --warmup 20 --precondition 100"
ER_OWN_ONLY=16 run only16 $Q
ER_OWN_ONLY=1 run only1 $Q
ER_OWN_NOPROJ=1 run noproj $Q
ER_OWN_ONLY=1 ER_OWN_NOPROJ=1 run only1_noproj $Q
ER_OWN_ONLY=16 ER_OWN_NOPROJ=1 run only16_noproj $Q
