#!/bin/bash
# round 5 session 7: the whole -m gpu suite on the round-5 library (hand-written device-wide sort + head scan instead of
# rocPRIM, row records, in-register catch-up, one-launch sort + lookup), then the bench lines that moved: DeepFM (and the
# 5-waves build of the own launch), DIN, MMoE, the embedding-parallel path over a world-1 RCCL group
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s7; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x --timeout 600 2>&1 | tail -6 | tee $O/tests.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step | parity', p.get('max_rel_loss_diff'), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])), '| emb', round((r.get('embedding_stage') or {}).get('us_per_step', 0), 1), round((r.get('embedding_stage') or {}).get('frac_of_hbm_peak', 0), 4))
print('   ', ' | '.join('%s %.1f/%s' % (k['kernel'][:26], k['us_per_step'], k['launches_per_step']) for k in r.get('kernels', []) if 'emb' in k['kernel'] or 'rocprim' in k['kernel']))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 2 --steady_steps 0 --steps 200 --warmup 20 --precondition 128"
run deepfm $Q
EASYREC_AMD_LIB=$PWD/_ab/w5/libeasyrec_hip.so run deepfm_w5 $Q
EASYREC_AMD_LIB=$PWD/_ab/t1p16/libeasyrec_hip.so run deepfm_t1p16 $Q
run din10m $Q --config configs/din_taobao_10m.config
run mmoe25m $Q --config configs/mmoe_taobao_4task_d64_25m.config
run ep1_rccl $Q --force_ep --rccl
EASYREC_AMD_EP_OVERLAP=0 run ep1_rccl_serial $Q --force_ep --rccl
run deepfm_again $Q
