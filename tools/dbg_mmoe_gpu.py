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
B = 128
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 23
est = EasyRecEstimator(cfg, device='cuda:0', batch_size=B, seed=SEED).build()
orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=SEED + 100)
b = gen.next_batch()
est.train_step(b); orc.train_step(b)
st = est.state_dict(slots=True)
rows = []
for k in orc.state:
  key = k + '/m'
  if key in orc.slots and key in st:
    ref = orc.slots[key]
    d, s = float(np.max(np.abs(st[key] - ref))), float(np.max(np.abs(ref)))
    rows.append((d / (s + 1e-30), k, d, s))
rows.sort(reverse=True)
for r in [r for r in rows if not r[1].endswith('/bias')][:3]:
  print('%.3e %-60s d=%.3e scale=%.3e' % r)
