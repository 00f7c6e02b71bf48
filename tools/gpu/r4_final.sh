#!/bin/bash
# end-of-round-4 evidence run: full -m gpu suite, smoke, the bench lines of BASELINE configs 2-5 (+ uniform ids, the
# embedding-parallel path over a world-1 RCCL group), rocprofv3 kernel stats of the default bench command, and the
# FETCH_SIZE / WRITE_SIZE counter passes (separate --pmc runs, --kernel-trace only) over the eager DeepFM step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04final; mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1; nproc >> $O/device.txt; lscpu | grep "Model name" >> $O/device.txt
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
run() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}; r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}; c=d.get('cpu_baseline') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], '| steady', round(s.get('ms_per_step_mean',0),4), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| cpu', c.get('value'), c.get('cores'))
print('   dom', (r.get('kernel') or '')[:60], r.get('achieved'), r.get('frac'), 'traffic', r.get('traffic'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
e=r.get('embedding_stage') or {}; g=r.get('gemm_family') or {}
print('   embedding stage', e.get('us_per_step'), e.get('frac_of_hbm_peak'), '| gemm family', g.get('TFLOPs'), g.get('frac_of_mfma_peak'))
" | tee -a $O/lines_summary.txt; }
echo default | tee -a $O/lines_summary.txt; run bench_default
echo dcnv2_f32 | tee -a $O/lines_summary.txt; run dcnv2_f32 --config configs/dcn_v2_criteo.config --steady_steps 256 --precondition 256 --cpu_seconds 3
echo dcnv2_bf16 | tee -a $O/lines_summary.txt; run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 --steady_steps 256 --precondition 256 --cpu_seconds 3
echo din10m | tee -a $O/lines_summary.txt; run din10m --config configs/din_taobao_10m.config --steady_steps 128 --precondition 128 --cpu_seconds 3
echo mmoe25m | tee -a $O/lines_summary.txt; run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --cpu_seconds 2
echo uniform | tee -a $O/lines_summary.txt; run uniform --ids uniform --no_cpu_baseline --steady_steps 256
echo ep1_rccl | tee -a $O/lines_summary.txt; run ep1_rccl --force_ep --rccl --no_cpu_baseline --steady_steps 0
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 > $O/prof.log 2>&1
ls -la $O/prof/*/ 2>/dev/null | head -12; cp $O/prof/bench_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null || cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_default.csv; rm -rf $O/prof  # (only the summary travels back: gpurun merges at most 64 MiB)
pass() { tag=$1; ctr=$2; shift 2; timeout 400 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | cut -c1-200; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 30 --warmup 5 --steady_steps 0 --precondition 160"
pass d_fs "FETCH_SIZE" $BENCH
pass d_ws "WRITE_SIZE" $BENCH
python - <<'PY' | tee $O/pmc_summary.txt
import csv, glob, collections, json
O='gpurun_out/r04final'
res={}
for tag in ('d_fs','d_ws'):
  agg=collections.defaultdict(lambda: collections.defaultdict(list))
  for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
    for r in csv.DictReader(open(f)):
      k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
      if 'gemm_f32_kernel' in k: k += ' grid=%s' % r.get('Grid_Size', r.get('Grid_Size_X', '?'))
      agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
  for k,c in agg.items():
    for name,v in c.items():
      v=v[len(v)//3:]  # drop pre-conditioning / warm-up launches
      d=res.setdefault('default',{}).setdefault(k,{})
      d[name]=sum(v)/max(len(v),1)
      d['launches']=len(v)
for k,c in sorted(res.get('default',{}).items(), key=lambda kv: -kv[1].get('FETCH_SIZE',0)):
  if 'er::' in k: print('%-72s'%k[:72], ' '.join('%s=%.4g'%(n,v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/pmc_by_kernel.json','w'), indent=1)
PY
rm -rf $O/d_fs $O/d_ws 2>/dev/null; ls $O; du -sh $O gpurun_out
