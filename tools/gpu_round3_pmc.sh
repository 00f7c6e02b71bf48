#!/bin/bash
# round 3: HBM-side traffic and VALU / VMEM activity counters per kernel of the DeepFM step (separate --pmc passes,
# --kernel-trace only), for the default step and for the exact step-by-step replay (EASYREC_AMD_EXACT_DECAY=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03pmc; mkdir -p $O
pass() { tag=$1; ctr=$2; shift 2; timeout 400 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; tail -2 $O/$tag.log; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 30 --warmup 5 --steady_steps 0 --precondition 160"
pass d_fs  "FETCH_SIZE" $BENCH
pass d_ws  "WRITE_SIZE" $BENCH
pass d_v1  "SQ_INSTS_VALU SQ_WAVE_CYCLES" $BENCH
pass d_v2  "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" $BENCH
pass d_v3  "SQ_BUSY_CYCLES SQ_WAVES" $BENCH
export EASYREC_AMD_EXACT_DECAY=1
pass x_v1  "SQ_INSTS_VALU SQ_WAVE_CYCLES" $BENCH
pass x_v2  "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" $BENCH
pass x_fs  "FETCH_SIZE" $BENCH
pass x_ws  "WRITE_SIZE" $BENCH
unset EASYREC_AMD_EXACT_DECAY
python - <<'PY' | tee $O/summary.txt
import csv, glob, collections, json
O='gpurun_out/r03pmc'
def load(tag):
  agg=collections.defaultdict(lambda: collections.defaultdict(list))
  for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
    for r in csv.DictReader(open(f)):
      k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
      # GEMM launches of one kernel differ by shape: key the forward kernel by its grid too
      if 'gemm_f32_kernel' in k: k += ' grid=%s' % r.get('Grid_Size', r.get('Grid_Size_X', '?'))
      agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
  return agg
res={}
for tag in ('d_fs','d_ws','d_v1','d_v2','d_v3','x_v1','x_v2','x_fs','x_ws'):
  a=load(tag)
  mode='default' if tag[0]=='d' else 'exact_replay'
  for k,c in a.items():
    for name,v in c.items():
      v=v[len(v)//3:]  # drop pre-conditioning / warm-up launches
      d=res.setdefault(mode,{}).setdefault(k,{})
      d[name]=sum(v)/max(len(v),1)
      d['launches']=len(v)
for mode in res:
  print('==',mode)
  for k,c in sorted(res[mode].items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get('FETCH_SIZE',0))):
    if k.startswith('er::') or 'er::' in k:
      print('%-72s'%k[:72], ' '.join('%s=%.4g'%(n,v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/pmc_by_kernel.json','w'), indent=1)
PY
rm -rf $O/*/ 2>/dev/null; ls $O
