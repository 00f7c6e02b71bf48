#!/bin/bash
# what the weight-gradient kernel of a batch-long contraction waits for: event timings per shape + PMC passes on the
# 128 x 128 x 204,800 problem (default kernel and natural-layout kernel in the same process)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
export ER_WGRAD_MAX_SPLITS=1024
timeout 300 python tools/wgrad_probe.py 2>&1 | tee $O/probe.log
for rows in 512 8192; do echo "rows/split $rows"; ER_WGRAD_SPLIT_ROWS=$rows timeout 300 python tools/wgrad_probe.py --only 0 2>&1 | tee -a $O/probe.log; done
pass() { tag=$1; ctr=$2; shift 2; timeout 300 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log; }
P="python tools/wgrad_probe.py --only 0 --iters 6"
pass p1 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" $P
pass p2 "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $P
pass p3 "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $P
pass p4 "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum" $P
pass p5 "GRBM_COUNT GRBM_GUI_ACTIVE FETCH_SIZE" $P
pass p6 "SQ_WAVES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES" $P
python - <<'PY' | tee $O/pmc_summary.txt
import csv, glob, collections
O='gpurun_out/r03m'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('p1','p2','p3','p4','p5','p6'):
  for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
    for r in csv.DictReader(open(f)):
      k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
      if 'gemm' not in k: continue
      agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in agg.items():
  print(k)
  for name,v in sorted(c.items()):
    v=v[len(v)//2:]
    print('    %-34s %14.0f  (n=%d)' % (name, sum(v)/len(v), len(v)))
PY
