#!/bin/bash
# round 4 session 29: rocprofv3 kernel stats of the default bench command on the final tree (after er_wide_fm_concat)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s29; mkdir -p $O
timeout 110 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 > $O/prof.log 2>&1
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null || cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_default.csv; rm -rf $O/prof
grep '^{' $O/prof.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms/step under rocprofv3')"
head -3 $O/kernel_stats_default.csv | cut -c1-120
