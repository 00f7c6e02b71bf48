#!/bin/bash
# PMC counters of er_gemm_bf16_nt on the DCN-v2 cross shape and 4096^3 (separate passes, --kernel-trace only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ntpmc; mkdir -p $O; rm -f $O/summary.txt
i=0
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace -f csv -d $O/p$i -o g -- python tools/gemm_bf16_bench.py nt_only > $O/p$i.log 2>&1
  python - "$O/p$i" <<'PY' >> $O/summary.txt
import csv,sys,collections,glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'gemm_bf16_nt' not in r['Kernel_Name']: continue
    agg[r.get('Grid_Size','')][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items(), key=lambda kv: float(kv[0] or 0)):
  print('grid', k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=%d'%len(next(iter(v.values()))))
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
