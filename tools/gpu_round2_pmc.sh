#!/bin/bash
# HBM-side request counters of the embedding kernels: separate passes, --kernel-trace only
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02pmc; mkdir -p $O
pass() { tag=$1; ctr=$2; shift 2; timeout 300 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 40 --warmup 5 --steady_steps 0 --precondition 256"
pass cal_rd  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" python tools/pmc_calibrate.py
pass cal_wr  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" python tools/pmc_calibrate.py
pass cal_fs  "FETCH_SIZE" python tools/pmc_calibrate.py
pass cal_ws  "WRITE_SIZE" python tools/pmc_calibrate.py
pass b_rd  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" $BENCH
pass b_wr  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $BENCH
pass b_fs  "FETCH_SIZE" $BENCH
pass b_ws  "WRITE_SIZE" $BENCH
python - <<'PY' | tee $O/summary.txt
import csv, glob, collections, json
O='gpurun_out/r02pmc'
def load(tag):
  agg=collections.defaultdict(lambda: collections.defaultdict(list))
  for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
    for r in csv.DictReader(open(f)):
      k=r['Kernel_Name'].split('(')[0].replace('void ','')[:48]
      agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
  return agg
res={}
for tag in ('cal_rd','cal_wr','cal_fs','cal_ws','b_rd','b_wr','b_fs','b_ws'):
  a=load(tag)
  for k,c in a.items():
    for name,v in c.items():
      v=v[len(v)//4:]  # drop warm-up launches
      res.setdefault(tag[:3].rstrip('_'),{}).setdefault(k,{})[name]=sum(v)/max(len(v),1)
      res[tag[:3].rstrip('_')][k]['n']=len(v)
for grp in res:
  print('==',grp)
  for k,c in sorted(res[grp].items()):
    if any(s in k for s in ('stream_copy','adam_decay_sweep','gather_rows','emb_','flush_window')):
      print('%-50s'%k, ' '.join('%s=%.0f'%(n,v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/pmc_summary.json','w'), indent=1)
PY
rm -rf $O/*/ 2>/dev/null; ls $O
