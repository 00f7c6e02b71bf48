#!/bin/bash
# round 6 session 6: BASELINE config 5 at its stated 200 M rows - parity against the compact-table oracle; wave-state
# counters (SQ passes, --kernel-trace only) for the embedding kernels of the eager DeepFM step; the counter list of the box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s6; mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1; grep -c . $O/counters_available.txt
line() { name=$1; shift; ( timeout 1500 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', json.dumps(p)[:700])
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
" | tee -a $O/lines_summary.txt; }
echo mmoe200m_parity | tee -a $O/lines_summary.txt; line mmoe200m --config configs/mmoe_taobao_4task_d64_200m.config --no_cpu_baseline --parity_only --steady_steps 0 --precondition 64 --steps 20 --warmup 5
tail -5 $O/mmoe200m.out | cut -c1-400
pass() { tag=$1; ctr="$2"; shift 2; timeout 400 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; echo "$tag exit $?"; tail -2 $O/$tag.log | cut -c1-200; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 30 --warmup 5 --steady_steps 0 --precondition 160 --parity_steps 0"
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" $BENCH
pass sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $BENCH
pass sq3 "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT SQ_IFETCH" $BENCH
python - <<'PY' | tee $O/sq_by_kernel.txt
import csv, glob, collections, json
O='gpurun_out/r6s6'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('sq1','sq2','sq3'):
  for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
    for r in csv.DictReader(open(f)):
      k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
      if 'gemm_f32_kernel' in k: k += ' grid=%s' % r.get('Grid_Size', r.get('Grid_Size_X', '?'))
      agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res={}
for k,c in agg.items():
  if 'er::' not in k: continue
  res[k]={n:sum(v[len(v)//3:])/max(len(v[len(v)//3:]),1) for n,v in c.items()}
  res[k]['launches']=max(len(v) for v in c.values())
for k,c in sorted(res.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES',0)):
  print('%-60s'%k[:60], ' '.join('%s=%.4g'%(n.replace('SQ_',''),v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/sq_by_kernel.json','w'), indent=1)
PY
rm -rf $O/sq1 $O/sq2 $O/sq3
ls $O; du -sh $O
