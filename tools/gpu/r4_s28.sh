#!/bin/bash
# round 4 session 28: last check of the final tree: device max F1 test, the embedding-parallel W = 1 RCCL line and the
# default line after er_wide_fm_concat
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s28; mkdir -p $O
timeout 60 python -m pytest tests/test_metric_pins.py tests/test_metrics.py -q -m gpu --timeout 50 2>&1 | tail -2 | tee $O/tests.log
( timeout 60 python bench.py --force_ep --rccl --no_cpu_baseline --steady_steps 0 --precondition 256 ) > $O/ep1_rccl.out 2>&1; grep '^{' $O/ep1_rccl.out | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ep1_rccl', round(d['ms_per_step'],4))" | tee -a $O/lines.log; grep -E "Error|Traceback" $O/ep1_rccl.out | head -3
( timeout 100 python bench.py --steady_steps 256 --cpu_seconds 3 ) > $O/bench_default.out 2>&1; grep '^{' $O/bench_default.out | tail -1 > $O/bench_default_line.jsonl; python -c "
import json; d=json.loads(open('$O/bench_default_line.jsonl').read()); r=d['roofline']; p=d.get('parity_full_size') or {}
print('default', round(d['ms_per_step'],4), 'ms', round(d['value']), 'ex/s | parity', p.get('max_rel_loss_diff'), '| launches', sum(f['launches_per_step'] for f in r['families']), '| emb stage', r['embedding_stage']['us_per_step'], r['embedding_stage']['frac_of_hbm_peak'])
" | tee -a $O/lines.log
