#!/bin/bash
# DIN's plain towers through run_parallel: DIN tests + DIN step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03ad; mkdir -p $O
timeout 300 python -m pytest tests -q -m gpu -x -k "din" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'ms/step | steady', round(s.get('ms_per_step_mean',0),4), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
run din --config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50
