#!/bin/bash
# round 4, session 8: PMC pass over the warm head / tail / ce chains: are the waves long-lived (in-kernel latency) or is it dispatch?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s8; mkdir -p $O
for op in head tail ce; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/pmc_$op -o p -- $R/tools/micro/lib_chain $op 8 4 > $O/pmc_$op.log 2>&1
  python3 - $O/pmc_$op <<'PY'
import sys, glob, csv, collections
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
  for row in csv.DictReader(open(f)):
    agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in agg.items():
  print(k, {n: round(sum(v) / len(v), 1) for n, v in c.items()})
PY
done 2>&1 | tee $O/pmc_summary.txt
