#!/bin/bash
# round 5 session 16: after the removal of the rejected variants (GemmArgs 320 -> 152 bytes): the whole -m gpu suite, then
# the lines of configs 2-5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s16; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -12 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| dom', (r.get('kernel') or '')[:40], round(r.get('us_per_step', 0), 1), round(r.get('frac') or 0, 4))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
run deepfm $Q
run deepfm_again $Q
for c in dcn_v2_criteo din_taobao_10m mmoe_taobao_4task_d64_25m; do run $c $Q --config configs/$c.config; done
run dcn_v2_bf16 $Q --config configs/dcn_v2_criteo.config --dense_dtype bf16
