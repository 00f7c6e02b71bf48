#!/bin/bash
# round 3: dense tail on a second stream (EASYREC_AMD_OVERLAP_DENSE) re-measured against the default, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| kernel time', r.get('kernel_time_us_per_step'))
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
run base --no_cpu_baseline --steady_steps 256
EASYREC_AMD_OVERLAP_DENSE=1 run overlap_dense --no_cpu_baseline --steady_steps 256
run base2 --no_cpu_baseline --steady_steps 256
