#!/bin/bash
# round 5 session 10: (a) the tail's contraction by the number of workgroups its k-splits aim at (EASYREC_AMD_TAIL_BLOCKS),
# with and without whole splits per XCD; (b) DIN / MMoE with ER_WGRAD_XCD on / off, same box; (c) HBM-side traffic of the
# stand-alone grouped weight-gradient launch with and without the XCD region (FETCH_SIZE / WRITE_SIZE, separate --pmc passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s10; mkdir -p $O
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:30], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if ('emb' in k['kernel'] or 'grouped' in k['kernel'] or 'dense_opt' in k['kernel'])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
for tb in 128 192 256 320 384 512; do EASYREC_AMD_TAIL_BLOCKS=$tb run tail_$tb $Q; done
ER_WGRAD_XCD=0 EASYREC_AMD_TAIL_BLOCKS=512 run tail_512_noxcd $Q
EASYREC_AMD_TAIL_BLOCKS=256 run tail_256_again $Q --parity_steps 2
run din10m_xcd $Q --config configs/din_taobao_10m.config
ER_WGRAD_XCD=0 run din10m_noxcd $Q --config configs/din_taobao_10m.config
run mmoe25m_xcd $Q --config configs/mmoe_taobao_4task_d64_25m.config
ER_WGRAD_XCD=0 run mmoe25m_noxcd $Q --config configs/mmoe_taobao_4task_d64_25m.config
pass() { tag=$1; ctr=$2; shift 2; timeout 300 rocprofv3 --pmc $ctr --kernel-trace -f csv -d $O/$tag -o p -- "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | cut -c1-200; }
BENCH="python bench.py --no_cpu_baseline --no_graph --steps 30 --warmup 5 --steady_steps 0 --precondition 64 --parity_steps 0"
export EASYREC_AMD_FUSED_TAIL=0
ER_WGRAD_XCD=1 pass x1_fs "FETCH_SIZE" $BENCH
ER_WGRAD_XCD=1 pass x1_ws "WRITE_SIZE" $BENCH
ER_WGRAD_XCD=0 pass x0_fs "FETCH_SIZE" $BENCH
ER_WGRAD_XCD=0 pass x0_ws "WRITE_SIZE" $BENCH
unset EASYREC_AMD_FUSED_TAIL
pass t_fs "FETCH_SIZE" $BENCH
pass t_ws "WRITE_SIZE" $BENCH
python - <<'PY' | tee $O/pmc_summary.txt
import csv, glob, collections, json
O='gpurun_out/r5s10'
res={}
for var in ('x1','x0','t'):
  for tag in (var+'_fs', var+'_ws'):
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
  print('== variant', var, '(x1: tail as four launches, splits by XCD; x0: the round-4 grid; t: the fused tail)')
  for k,c in sorted(res[var].items(), key=lambda kv: -kv[1].get('FETCH_SIZE',0)):
    if 'er::' in k and ('grouped' in k or 'emb_bwd' in k or 'splitk' in k): print('%-72s'%k[:72], ' '.join('%s=%.4g'%(n,v) for n,v in sorted(c.items())))
json.dump(res, open(O+'/pmc_by_kernel.json','w'), indent=1)
PY
rm -rf $O/x1_fs $O/x1_ws $O/x0_fs $O/x0_ws $O/t_fs $O/t_ws 2>/dev/null; du -sh $O
