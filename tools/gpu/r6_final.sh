#!/bin/bash
# end-of-round-6 evidence run: smoke, the default bench line (steady state, parity, cpu_baseline, from-file rate), rocprofv3
# kernel stats of the default command, the FETCH_SIZE / WRITE_SIZE counter passes (separate --pmc runs, --kernel-trace only)
# over the eager DeepFM step and over the eager DCN-v2 bf16 step, the lines of BASELINE configs 3-5 (+ uniform ids, EP at
# W = 1, config 5 at its stated 200 M rows with the compact-oracle parity), the step timeline, the whole GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06final; mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1; nproc >> $O/device.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
run() { name=$1; shift; ( timeout 1500 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}; r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}; c=d.get('cpu_baseline') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], '| steady', round(s.get('ms_per_step_mean',0),4), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), p.get('oracle_tables'), '| cpu', c.get('value'), c.get('cores'), '| clocks', d.get('clocks'))
print('   dom', (r.get('kernel') or '')[:60], r.get('achieved'), r.get('frac'), 'traffic', r.get('traffic'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
e=r.get('embedding_stage') or {}; g=r.get('gemm_family') or {}; u=(r.get('tail_unfused') or {}).get('embedding_stage_alone') or {}
print('   embedding stage', e.get('us_per_step'), e.get('frac_of_hbm_peak'), '| alone', u.get('us_per_step'), u.get('frac_of_hbm_peak'), '| gemm family', g.get('TFLOPs'), g.get('frac_of_mfma_peak'))
ff=d.get('from_file')
if ff: print('   from_file', json.dumps(ff)[:500])
" | tee -a $O/lines_summary.txt; }
echo default | tee -a $O/lines_summary.txt; run bench_default
echo default_from_csv | tee -a $O/lines_summary.txt; run bench_default_csv --no_cpu_baseline --steady_steps 0 --from_file csv
echo default_from_criteo | tee -a $O/lines_summary.txt; run bench_default_criteo --no_cpu_baseline --steady_steps 0 --from_file criteo
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 --parity_steps 0 > $O/prof.log 2>&1
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null || cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_default.csv; rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/proft -o trace -- python bench.py --steps 200 --warmup 20 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 64 > $O/proft.log 2>&1
DB=$(find $O/proft -name "*.db" | head -1); python tools/rocpd_timeline.py $DB 100 > $O/default_step_timeline.txt 2>&1; tail -2 $O/default_step_timeline.txt | cut -c1-160; rm -rf $O/proft
pass() { tag=$1; ctr=$2; shift 2; timeout 400 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | cut -c1-160; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 30 --warmup 5 --steady_steps 0 --precondition 160 --parity_steps 0"
pass d_fs "FETCH_SIZE" $BENCH
pass d_ws "WRITE_SIZE" $BENCH
pass b_fs "FETCH_SIZE" $BENCH --config configs/dcn_v2_criteo.config --dense_dtype bf16
pass b_ws "WRITE_SIZE" $BENCH --config configs/dcn_v2_criteo.config --dense_dtype bf16
python - <<'PY' | tee $O/pmc_summary.txt
import csv, glob, collections, json
O='gpurun_out/r06final'
res={}
for cfg,tags in (('default',('d_fs','d_ws')),('dcnv2_bf16',('b_fs','b_ws'))):
  for tag in tags:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
      for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
        if 'gemm_f32_kernel' in k: k += ' grid=%s' % r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
      for name,v in c.items():
        v=v[len(v)//3:]
        d=res.setdefault(cfg,{}).setdefault(k,{})
        d[name]=sum(v)/max(len(v),1)
        d['launches']=len(v)
for cfg in res:
  print(cfg)
  for k,c in sorted(res[cfg].items(), key=lambda kv: -kv[1].get('FETCH_SIZE',0))[:24]:
    if 'er::' in k: print('  %-72s'%k[:72], ' '.join('%s=%.4g'%(n,v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/pmc_by_kernel.json','w'), indent=1)
PY
rm -rf $O/d_fs $O/d_ws $O/b_fs $O/b_ws 2>/dev/null
Q="--steady_steps 256 --precondition 256 --cpu_seconds 2"
echo dcnv2_f32 | tee -a $O/lines_summary.txt; run dcnv2_f32 --config configs/dcn_v2_criteo.config $Q
echo dcnv2_bf16 | tee -a $O/lines_summary.txt; run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 $Q
echo din10m | tee -a $O/lines_summary.txt; run din10m --config configs/din_taobao_10m.config --steady_steps 128 --precondition 128 --cpu_seconds 2
echo mmoe25m | tee -a $O/lines_summary.txt; run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --cpu_seconds 2
echo mmoe200m_compact_parity | tee -a $O/lines_summary.txt; run mmoe200m --config configs/mmoe_taobao_4task_d64_200m.config --no_cpu_baseline --parity_only --steady_steps 0 --precondition 64
echo uniform | tee -a $O/lines_summary.txt; run uniform --ids uniform --no_cpu_baseline --steady_steps 256
echo ep1_rccl | tee -a $O/lines_summary.txt; run ep1_rccl --force_ep --rccl --no_cpu_baseline --steady_steps 0
for cfg in din_taobao_10m mmoe_taobao_4task_d64_25m; do
  timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/proft_$cfg -o trace -- python bench.py --config configs/$cfg.config --steps 100 --warmup 20 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 32 > $O/proft_$cfg.log 2>&1
  DB=$(find $O/proft_$cfg -name "*.db" | head -1); python tools/rocpd_timeline.py $DB 60 > $O/${cfg}_step_timeline.txt 2>&1; tail -1 $O/${cfg}_step_timeline.txt | cut -c1-160; rm -rf $O/proft_$cfg
done
timeout 2400 python -m pytest tests -q --timeout 900 -m gpu 2>&1 | tail -12 | tee $O/tests_full.txt
ls $O; du -sh $O
