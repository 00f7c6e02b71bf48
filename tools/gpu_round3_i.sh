#!/bin/bash
# multi-layer BatchNorm launches (EASYREC_AMD_GROUPED_BN) A/B on MMoE 25 M + the tests that hold them to the single launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -m gpu -x -k "multi_layer or grouped or mmoe or multi_task" 2>&1 | tail -15 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| parity', (d.get('parity_full_size') or {}).get('max_rel_err'))
for k,v in sorted((r.get('families') or {}).items(), key=lambda kv:-kv[1].get('us',0)): print('   ', k, v.get('us'), v.get('launches'))
"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --cpu_seconds 2"
run mmoe_grouped_bn $M
EASYREC_AMD_GROUPED_BN=0 run mmoe_single_bn $M
run deepfm --no_cpu_baseline --steady_steps 256
