#!/bin/bash
# round 5 session 1: row records (var | m | v | last_step in one record per row, er_emb_group_set_row_pitch): the whole -m gpu
# suite on the record layout, then a same-box A/B of the DeepFM line against plain [rows, dim] arrays
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s1; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x --timeout 300 2>&1 | tail -6 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step |', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', (r.get('embedding_stage') or {}))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
EASYREC_AMD_ROW_RECORDS=0 run planar $Q
run records $Q
EASYREC_AMD_ROW_RECORDS=0 run planar_again $Q
run records_again $Q
