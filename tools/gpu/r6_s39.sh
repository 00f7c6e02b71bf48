#!/bin/bash
# round 6 session 39: rocprofv3 --kernel-trace --stats tables of the DIN, MMoE and DCN-v2 (fp32 / bf16) commands on the final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s39; mkdir -p $O
for c in "din10m --config configs/din_taobao_10m.config" "mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config" "dcnv2_f32 --config configs/dcn_v2_criteo.config" "dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16"; do
  set -- $c; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$name -o bench -- python bench.py "$@" --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 32 > $O/prof_$name.log 2>&1
  cp $O/prof_$name/bench_kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null || cp $O/prof_$name/*/*kernel_stats.csv $O/kernel_stats_$name.csv
  rm -rf $O/prof_$name; grep '^{' $O/prof_$name.log | tail -1 | cut -c1-200; head -4 $O/kernel_stats_$name.csv | cut -c1-160
done
