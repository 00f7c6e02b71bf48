#!/bin/bash
# round 3, second GPU call: deferred BatchNorm (A transform in the GEMM staging, finalize by the last workgroup) - bit
# equality tests, the pins on the HIP kernels, A/B bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k 'deferred or gemm or batchnorm or closed_form or bn_' 2>&1 | tail -15 | tee $O/kernel_tests.txt
timeout 600 python -m pytest tests/test_deepfm_gpu.py -m gpu -q -s -k 'deferred or closed_form or evaluate_does or fused_batchnorm' 2>&1 | tail -15 | tee $O/model_tests.txt
timeout 300 python -m pytest tests/test_embedding_stage_pins.py -m gpu -q 2>&1 | tail -15 | tee $O/pins.txt
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| dom', r.get('kernel'), r.get('us_per_step'), r.get('frac'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))
for f in r.get('families', []): print('   ', f['family'], round(f['us_per_step'],1), round(f['share'],3), f['launches_per_step'])
for k in r.get('kernels', [])[:14]: print('      ', round(k['us_per_step'],1), k['launches_per_step'], k['kernel'][:80])
print('   emb stage', (r.get('embedding_stage') or {}).get('GBps'), (r.get('embedding_stage') or {}).get('frac_of_hbm_peak'), 'gemm family', r.get('gemm_family'))
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
run default --no_cpu_baseline --steady_steps 512
EASYREC_AMD_DEFER_BN=0 run nodefer --no_cpu_baseline --steady_steps 512
run din10m --config configs/din_taobao_10m.config --no_cpu_baseline --steady_steps 128 --precondition 128
EASYREC_AMD_DEFER_BN=0 run din10m_nodefer --config configs/din_taobao_10m.config --no_cpu_baseline --steady_steps 128 --precondition 128
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128
