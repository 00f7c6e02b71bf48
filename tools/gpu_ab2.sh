#!/bin/bash
# same-box A/B: _ab/base (exported commit, tools/ab_export.sh) against the working tree.  usage: gpu_ab2.sh "<bench args>" [reps]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ARGS="$1"; REPS=${2:-3}
ms() { grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'steady', round(s.get('ms_per_step_mean',0),4), 'catch_up', round(s.get('catch_up_ms_p50',0),4))"; }
for rep in $(seq $REPS); do
  echo "base $(cat _ab/base/AB_COMMIT): $(cd _ab/base && EASYREC_AMD_FUSED_BN_GEMM=0 timeout 600 python bench.py --no_cpu_baseline $ARGS 2>&1 | ms)"
  echo "head: $(timeout 600 python bench.py --no_cpu_baseline $ARGS 2>&1 | ms)"
done
