#!/bin/bash
# round 5 session 11: XCD placement generalised (4 / 2 splits on 2 / 4 XCDs) and restricted to problems whose split count is
# target-driven: grouped GEMM tests, DeepFM default (tail at 256 = 4 splits), DIN / MMoE with ER_WGRAD_XCD on / off same box,
# FETCH_SIZE / WRITE_SIZE of the fused tail launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s11; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py -q -m gpu --timeout 300 -k "gemm_grouped or fused_step_variants or wgrad or grouped" 2>&1 | tail -5 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:30], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if ('emb' in k['kernel'] or 'grouped' in k['kernel'] or 'dense_opt' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
run deepfm $Q --parity_steps 2
ER_WGRAD_XCD=0 run deepfm_noxcd $Q
run deepfm_again $Q
run din10m_xcd $Q --config configs/din_taobao_10m.config
ER_WGRAD_XCD=0 run din10m_noxcd $Q --config configs/din_taobao_10m.config
run mmoe25m_xcd $Q --config configs/mmoe_taobao_4task_d64_25m.config
ER_WGRAD_XCD=0 run mmoe25m_noxcd $Q --config configs/mmoe_taobao_4task_d64_25m.config
run dcnv2 $Q --config configs/dcn_v2_criteo.config
ER_WGRAD_XCD=0 run dcnv2_noxcd $Q --config configs/dcn_v2_criteo.config
pass() { tag=$1; ctr=$2; shift 2; timeout 300 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | cut -c1-200; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 30 --warmup 5 --steady_steps 0 --precondition 64 --parity_steps 0"
pass t_fs "FETCH_SIZE" $BENCH
pass t_ws "WRITE_SIZE" $BENCH
ER_WGRAD_XCD=0 pass t0_fs "FETCH_SIZE" $BENCH
python - <<'PY' | tee $O/pmc_summary.txt
import csv, glob, collections, json
O='gpurun_out/r5s11'
res={}
for tag in ('t_fs','t_ws','t0_fs'):
    var = tag.split('_')[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/**/*counter_collection.csv'%(O,tag), recursive=True):
      for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
      for name,v in c.items():
        v=v[len(v)//3:]
        d=res.setdefault(var,{}).setdefault(k,{})
        d[name]=sum(v)/max(len(v),1)
        d['launches']=len(v)
for var in res:
  print('== variant', var, '(t: the fused tail, 4 splits on XCD pairs; t0: the same with ER_WGRAD_XCD=0)')
  for k,c in sorted(res[var].items(), key=lambda kv: -kv[1].get('FETCH_SIZE',0)):
    if 'er::' in k: print('%-72s'%k[:72], ' '.join('%s=%.4g'%(n,v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/pmc_by_kernel.json','w'), indent=1)
PY
rm -rf $O/t_fs $O/t_ws $O/t0_fs 2>/dev/null; du -sh $O
