#!/bin/bash
# full GPU regression + headline bench lines (used after every kernel change)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/all; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_all.log 2>&1; echo "exit $?" >> $O/pytest_all.log; tail -6 $O/pytest_all.log
for args in "" "--ids uniform" "--dense_sweep" "--optimizer lazy_adam" "--force_ep"; do
  echo "== bench $args"; timeout 600 python bench.py --no_cpu_baseline --steps 60 $args > $O/b.log 2>&1; tail -1 $O/b.log | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print(round(d['ms_per_step'],3),'ms', round(d['value']),'ex/s', d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))
except Exception as e: print('PARSE', l[-300:])
"
done
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline > $O/prof.log 2>&1
python tools/trace_summary.py $O/prof/bench_kernel_trace.csv > $O/all_by_shape.txt
rm -f $O/prof/*kernel_trace.csv
