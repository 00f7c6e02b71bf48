import sys, os, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.WARNING)
import numpy as np, torch
from easyrec_amd.input.synthetic import SyntheticBatches
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
from easyrec_amd.utils import config_util
from oracle.model_oracle import OracleTrainer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs/mmoe_taobao_small.config'))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 23
DSEED = int(sys.argv[3]) if len(sys.argv) > 3 else SEED + 100
est = EasyRecEstimator(cfg, device='cuda:0', batch_size=B, seed=SEED).build()
if len(sys.argv) > 4:  # start from the CPU-built state, as tests/test_golden_models.py does
  sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
  from make_model_vectors import initial_state
  est.load_state_dict(initial_state('mmoe_taobao_small.config', B, SEED)[2])
orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=DSEED)
b = gen.next_batch()
est.train_step(b); ol1 = orc.train_step(b)
st = est.state_dict(slots=True)
rows = []
for k in orc.state:
  key = k + '/m'
  if key in orc.slots and key in st:
    ref = orc.slots[key]
    d, s = float(np.max(np.abs(st[key] - ref))), float(np.max(np.abs(ref)))
    rows.append((d / (s + 1e-30), k, d, s))
rows.sort(reverse=True)
for r in [r for r in rows if r[3] > 1e-5][:14]:
  print('%.3e %-60s d=%.3e scale=%.3e' % r)

print('losses hip', est.loss_values())
print('losses orc', ol1)
b2 = gen.next_batch()
est.train_step(b2); ol2 = orc.train_step(b2)
print('step2 hip', est.loss_values())
print('step2 orc', ol2)
