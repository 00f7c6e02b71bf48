#!/bin/bash
# round 3, first GPU call: closed-form decay replay - parity canaries, default bench line, A/B against the exact replay,
# kernel stats, torch.profiler probe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k 'closed_form or lazy_dense' 2>&1 | tail -5 | tee $O/canary.txt
timeout 600 python -m pytest tests/test_deepfm_gpu.py -m gpu -q -s -k 'closed_form or lazy_decay_equals or evaluate_does or full_size' 2>&1 | tail -8 | tee $O/model_tests.txt
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}; p=d.get('parity_full_size') or {}; e=d.get('embedding_stage') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), 'p99', round(s.get('ms_per_step_p99',0),4), 'catch_up', round(s.get('catch_up_ms_p50',0),4), 'flush', round(s.get('flush_decay_ms',0),2), '|', d['dtype'], r.get('kernel'), round(r.get('achieved',0),1), r.get('unit'), 'frac', round(r.get('frac',0),3), '| cpu', c.get('value'), c.get('cores'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| stage GBps', e.get('stage_GBps'), e.get('stage_frac_of_hbm_peak'))"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
run default
EASYREC_AMD_EXACT_DECAY=1 run exact --no_cpu_baseline --steady_steps 512
run uniform --ids uniform --no_cpu_baseline --steady_steps 512
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 1024 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_default.csv --steps 2044 | tail -60 > $O/stats.txt; rm -rf $O/prof; tail -3 $O/stats.txt
timeout 200 python tools/profiler_probe.py > $O/profiler_probe.txt 2>&1; tail -30 $O/profiler_probe.txt
