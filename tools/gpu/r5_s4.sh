#!/bin/bash
# round 5 session 4: (1) the fused step's variants bit for bit + the stale-table detection (new tests); (2) sort + lookup
# in ONE launch with the replay table from the prologue (er_emb_front_fwd), one-row tables first in the own launch, 8
# composites per sort thread: same-box A/Bs; (3) per-workgroup stamps of the own launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s4; mkdir -p $O
timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_files_to_gpu.py tests/test_kv_embedding.py -q -m gpu -x --timeout 300 2>&1 | tail -6 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', (r.get('embedding_stage') or {}).get('us_per_step'), (r.get('embedding_stage') or {}).get('frac_of_hbm_peak'))
print('   ', ' | '.join('%s %.1f' % (k['kernel'][:28], k['us_per_step']) for k in r.get('kernels', []) if 'emb' in k['kernel'] or 'hyper' in k['kernel']))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
EASYREC_AMD_PROLOGUE_TABLES=0 EASYREC_AMD_PROJ_FIRST=0 run two_launches_proj_last $Q
EASYREC_AMD_PROLOGUE_TABLES=0 run two_launches_proj_first $Q
run one_launch $Q
EASYREC_AMD_SORT_E8=1 run one_launch_e8 $Q
EASYREC_AMD_PROLOGUE_TABLES=0 EASYREC_AMD_SORT_E8=1 run two_launches_e8 $Q
run one_launch_again $Q
EASYREC_AMD_PROJ_FIRST=0 timeout 200 python tools/own_probe.py 2>&1 | tail -8 | tee $O/own_probe_proj_last.txt
timeout 200 python tools/own_probe.py 2>&1 | tail -8 | tee $O/own_probe_proj_first.txt
EASYREC_AMD_PAIR_TILES=0 EASYREC_AMD_PROJ_FIRST=0 timeout 200 python tools/own_probe.py 2>&1 | tail -8 | tee $O/own_probe_unpaired.txt
