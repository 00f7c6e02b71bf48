"""Where do the closed-form and the exact replay part ways under embedding parallelism?  Per step: the largest relative
deviation of any table / slot between two runs that differ only in EASYREC_AMD_EXACT_DECAY, single GPU, EP W=1, EP W=2."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _sim_comm import SimWorld  # noqa: E402
from easyrec_amd.input.criteo_synthetic import SyntheticCriteo  # noqa: E402
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator  # noqa: E402
from easyrec_amd.model.embedding_parallel import EmbeddingParallelEstimator  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402

cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
B, STEPS = 64, 12
feats = list(cfg.feature_config.features)


def batches(seed):
  g = SyntheticCriteo(cfg.data_config, feats, batch_size=B, seed=seed)
  a = g.next_batch()
  ring = [g.next_batch() for _ in range(3)]
  return [a] + [ring[i % 3] for i in range(STEPS - 2)] + [a]


def dev(sa, sb):
  worst = (0.0, '')
  for k in sa:
    if sa[k].dtype.kind != 'f' or sa[k].size == 0:
      continue
    sc = max(float(np.abs(sb[k]).max()), 1e-30)
    e = float(np.abs(sa[k].astype(np.float64) - sb[k]).max()) / sc
    worst = (e, k) if e > worst[0] else worst
  return worst


def run(kind, world, exact, snap_every=3):
  os.environ['EASYREC_AMD_EXACT_DECAY'] = '1' if exact else '0'
  if kind == 'single':
    est = EasyRecEstimator(cfg, device='cuda:0', batch_size=B, seed=4).build()
    out = []
    for i, b in enumerate(batches(1)):
      est.train_step(b)
      if i % snap_every == snap_every - 1 or i == STEPS - 1:
        out.append(est.state_dict(slots=True))
    return out
  sim = SimWorld(world)
  scheds = [batches(1 + 10 * r) for r in range(world)]

  def rank_fn(rank, comm):
    torch.cuda.set_device(0)
    est = EmbeddingParallelEstimator(cfg, device='cuda:0', batch_size=B, seed=4, rank=rank, world=world, comm=comm,
                                     replicate_bytes=1024).build()
    out = []
    for i, b in enumerate(scheds[rank]):
      est.train_step(b)
      if i % snap_every == snap_every - 1 or i == STEPS - 1:
        out.append(est.state_dict(slots=True))
    return out

  return sim.run(rank_fn)[0]


for env in ({}, {'EASYREC_AMD_PADDED_EXCHANGE': '0'}, {'EASYREC_AMD_PADDED_EXCHANGE': '0', 'EASYREC_AMD_OWNER_MERGE': '0'},
            {'EASYREC_AMD_SHARE_ROUTE': '0'}):
  os.environ.update(env)
  a, b = run('ep', 2, False), run('ep', 2, True)
  print('ep 2', env, ['%.2e %s' % (d, k[-40:]) for d, k in (dev(x, y) for x, y in zip(a, b))])
  a2 = run('ep', 2, False)
  print('   closed twice:', ['%.2e' % dev(x, y)[0] for x, y in zip(a, a2)])
  for k in env:
    os.environ.pop(k)
