#!/bin/bash
# round 3, fifth GPU call: the re-shaped closed-form tests, the lock-step parallel stacks (MMoE) A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_embedding_parallel_gpu.py tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_golden_models.py -m gpu -q -s -k 'closed or deferred or evaluate_does or mmoe or grouped or ple or dbmtl or multi' 2>&1 | grep -v "^WARNING\|^$" | tail -40 | tee $O/tests.txt
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| dom', (r.get('kernel') or '')[:60], r.get('us_per_step'), r.get('frac'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), p.get('error'))
for f in r.get('families', []): print('   ', f['family'], round(f['us_per_step'],1), round(f['share'],3), f['launches_per_step'])
for k in r.get('kernels', [])[:14]: print('      ', round(k['us_per_step'],1), k['launches_per_step'], k['kernel'][:80])
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 1200 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
run mmoe25m_grouped --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128
EASYREC_AMD_GROUPED_STACKS=0 run mmoe25m_sequential --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128
