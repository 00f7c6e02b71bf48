#!/bin/bash
# round 5 session 5: the paired tiles for real (the dim-16 group carries the dim-1 group's tiles whichever of them leads the
# sort): variants bit for bit, same-box A/B (pairing, one-row tables first), per-workgroup stamps, per-kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s5; mkdir -p $O
timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_kernels_gpu.py -q -m gpu -x --timeout 300 -k "fused or variants or replay_table or first_step or trajectory or graph or emb" 2>&1 | tail -4 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p, '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', (r.get('embedding_stage') or {}).get('us_per_step'), (r.get('embedding_stage') or {}).get('frac_of_hbm_peak'))
print('   ', ' | '.join('%s %.1f' % (k['kernel'][:28], k['us_per_step']) for k in r.get('kernels', []) if 'emb' in k['kernel'] or 'hyper' in k['kernel']))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 300 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 300 --warmup 20 --precondition 256"
EASYREC_AMD_PAIR_TILES=0 EASYREC_AMD_PROJ_FIRST=0 run unpaired_proj_last $Q
EASYREC_AMD_PROJ_FIRST=0 run paired_proj_last $Q
run paired_proj_first $Q
EASYREC_AMD_PAIR_TILES=0 EASYREC_AMD_PROJ_FIRST=0 run unpaired_proj_last_again $Q
run paired_proj_first_again $Q
run paired_uniform $Q --ids uniform
EASYREC_AMD_PROJ_FIRST=0 timeout 200 python tools/own_probe.py 2>&1 | tail -6 | tee $O/own_probe_paired_proj_last.txt
timeout 200 python tools/own_probe.py 2>&1 | tail -6 | tee $O/own_probe_paired_proj_first.txt
