#!/bin/bash
# round 5 session 21: the hyper table refreshed without draining the stream (pinned staging + stream-ordered upload): the
# default line's steady state (2048 distinct batches cross one refresh), graph / restore tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s21; mkdir -p $O
timeout 600 python -m pytest tests/test_deepfm_gpu.py tests/test_files_to_gpu.py -q -m gpu --timeout 300 -k "graph or restore or continue or trajectory or deterministic" 2>&1 | tail -3 | tee $O/tests.log
( timeout 600 python bench.py --no_cpu_baseline --parity_steps 2 --steady_steps 6144 ) > $O/bench.out 2>&1; grep '^{' $O/bench.out | tail -1 > $O/bench_line.json
python -c "
import json
d=json.load(open('$O/bench_line.json')); s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'ms/step | steady', {k: (round(v,4) if isinstance(v,float) else v) for k,v in s.items() if 'ms_per' in k}, '| parity', (d.get('parity_full_size') or {}).get('max_rel_loss_diff'))
" | tee $O/line.txt
grep -E "Error|Traceback" $O/bench.out | head -3
