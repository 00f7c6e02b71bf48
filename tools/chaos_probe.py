"""How far apart do two runs of deepfm_criteo_small end up 5 steps after a one-ulp perturbation of the embeddings?  (CPU stand-in
backend; why tests/test_deepfm_gpu.py::_assert_closed_tracks_sweep bounds the bulk of the deviations, not their maximum.)"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from easyrec_amd import kernels
from oracle.kernel_ref import RefBackend
kernels._BACKEND = RefBackend()
from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
from easyrec_amd.utils import config_util
cfg = config_util.get_configs_from_pipeline_file('/root/repo/configs/deepfm_criteo_small.config')
B = 64
a = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=5).build()
b = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=5).build()
gen = SyntheticCriteo(cfg.data_config, a.feature_configs, batch_size=B, seed=8)
batches = [gen.next_batch() for _ in range(8)]
for i, bt in enumerate(batches):
  a.train_step(bt); b.train_step(bt)
  if i == 2:
    for dim, st in b.engine.storage.items():
      v = st['var']
      v.mul_(1.0 + 2e-7 * torch.sign(torch.randn_like(v)))  # a relative perturbation of one ulp-ish
sa, sb = a.state_dict(slots=True), b.state_dict(slots=True)
worst = {}
for k in sa:
  if sa[k].dtype.kind != 'f' or sa[k].size == 0: continue
  cls = 'm' if k.endswith('/m') else 'v' if k.endswith('/v') else 'var'
  scale = max(float(np.abs(sa[k]).max()), 1e-30)
  e = float(np.abs(sa[k].astype(np.float64) - sb[k]).max()) / scale
  if e > worst.get(cls, (0, None))[0]: worst[cls] = (e, k)
print(worst)
