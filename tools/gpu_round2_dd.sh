#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ee; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
timeout 300 python -m pytest tests/test_kv_embedding.py tests/test_reference_layers.py tests/test_models_gpu.py -m gpu -q --tb=short --timeout 150 2>&1 | grep -E "passed|failed|^E  |FAILED" | head -20
( time timeout 900 python -m pytest tests/test_kv_embedding.py tests/test_deepfm_gpu.py -m gpu -q --tb=line --timeout 200 2>&1 | tail -4 ) 2>&1 | grep -E "passed|failed|real|FAILED" | tee $O/suite.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), '| loss', d.get('final_loss'))"; }
python - <<'PY'
# a DeepFM-Criteo config whose 26 hashed features are hash-table backed (4 M-row arenas), for a bench line
import sys; sys.path.insert(0, '.')
from easyrec_amd.utils import config_util
cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo.config')
cfg.model_config.ev_params.max_capacity = 1 << 20
config_util.save_pipeline_config(cfg, 'gpurun_out/r02ee/deepfm_criteo_kv.config')
PY
echo "--- deepfm kv (1 M-row arenas)" | tee -a $O/lines.log
timeout 300 python bench.py --no_cpu_baseline --config $O/deepfm_criteo_kv.config --steady_steps 256 --precondition 256 > $O/kv.out 2>&1; grep '^{' $O/kv.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" -A3 $O/kv.out | head -8
