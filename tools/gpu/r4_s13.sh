#!/bin/bash
# round 4, session 13: PMC passes over the fused embedding kernels inside the (eager) DeepFM step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s13; mkdir -p $O
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -f csv -d $O/$name -o p -- python $R/bench.py --no_graph --steps 12 --warmup 3 --precondition 40 --no_cpu_baseline --parity_steps 0 --steady_steps 0 > $O/$name.log 2>&1
python3 - $O/$name <<'PY'
import sys, glob, csv, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
  for row in csv.DictReader(open(f)):
    k = row['Kernel_Name']
    if any(t in k for t in ('emb_bwd_own', 'emb_catch_up_heads', 'emb_front_sort', 'emb_fwd_kernel', 'emb_bwd_tile', 'catch_up_closed')):
      agg[k[:44]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in agg.items():
  print(k, {n: round(sum(v[len(v)//3:]) / max(len(v[len(v)//3:]), 1), 1) for n, v in c.items()})
PY
}
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM 2>&1 | tee $O/pmc_summary.txt
pass p2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_GDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS 2>&1 | tee -a $O/pmc_summary.txt
EASYREC_AMD_FUSED_EMB=0 pass p3 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM 2>&1 | tee -a $O/pmc_summary.txt
rm -rf $O/p1 $O/p2 $O/p3
