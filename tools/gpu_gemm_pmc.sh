#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/gemm_pmc; mkdir -p $O
i=0
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace -f csv -d $O/p$i -o g -- python tools/gemm_bench.py 0,1,2 f32 > $O/p$i.log 2>&1
  python - "$O/p$i/g_counter_collection.csv" <<'PY' >> $O/summary.txt
import csv,sys,collections
try:
  rows=list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
  print('no counters',e); sys.exit()
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
  if 'gemm_f32' not in r['Kernel_Name']: continue
  key=(r['Kernel_Name'][:45], r.get('Grid_Size',''))
  agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
  print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=%d'%len(next(iter(v.values()))))
PY
  rm -rf $O/p$i/*kernel_trace* 
done
cat $O/summary.txt
