#!/bin/bash
# round 5 session 23: last check of the final tree: smoke + the default line without the CPU legs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s23; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200 | tee $O/smoke.txt
( timeout 120 python bench.py --no_cpu_baseline --parity_steps 0 --steady_steps 0 ) 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], '| dom', (r.get('kernel') or '')[:40], round(r.get('frac') or 0,4), '| launches', sum(f['launches_per_step'] for f in r.get('families', [])))" | tee $O/line.txt
