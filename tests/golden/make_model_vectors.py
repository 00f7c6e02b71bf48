#!/usr/bin/env python
"""Writes tests/golden/model_vectors.json: first-training-step outputs of the model-level ORACLE
(oracle/model_oracle.py) for every hot-path model on seeded synthetic batches.

TensorFlow cannot run in this environment and the reference's tests hold no numeric expectation for these
graphs, so these are oracle outputs ("parity unpinned"): they (a) freeze the oracle against
accidental drift, (b) are what the GPU path is compared with on the GPU box, and (c) let a future session with a
TF install diff `python -m easy_rec.python.train_eval` against them.  Initial parameters are the product's
deterministic initialisation (seed recorded per case); batches come from the seeded generators.

  python tests/golden/make_model_vectors.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)

# Seeds are chosen so that no ReLU input of the two recorded steps lies within 5e-7 (of its column's largest) of zero (the ~0 outputs of
# batch-constant columns aside; main() asserts it): at a near-tie the ReLU mask - hence the gradient - depends on the summation order
# of the preceding GEMM, and a GPU run could not be compared with the CPU oracle beyond the first step.
CASES = [  # (config, batch size, estimator seed, data seed)
    ('deepfm_criteo_small.config', 64, 3, 101),
    ('dcn_criteo_small.config', 64, 3, 102),
    ('dcn_v2_criteo_small.config', 64, 3, 103),
    ('dcn_v2_lowrank_criteo_small.config', 64, 3, 104),
    ('din_taobao_small.config', 48, 3, 105),
    ('mmoe_taobao_small.config', 48, 4, 106),
]


def initial_state(config, B, seed):
  """The product's deterministic initial parameters, built on the CPU (embedding tables are drawn from a
  torch.Generator, whose CPU and GPU streams differ: the fixture is tied to the CPU stream)."""
  from easyrec_amd import kernels
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  from oracle.kernel_ref import RefBackend
  prev = kernels._BACKEND
  kernels._BACKEND = RefBackend()  # only to let the product build its initial state without a GPU
  try:
    cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
    est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=seed).build()
    return cfg, est.feature_configs, est.state_dict()
  finally:
    kernels._BACKEND = prev


def run_case(config, B, seed, data_seed):
  """Returns the dict stored in the fixture (also used by the tests to re-run the oracle)."""
  from easyrec_amd.input.synthetic import SyntheticBatches
  from oracle.model_oracle import OracleTrainer
  cfg, feature_configs, state = initial_state(config, B, seed)
  orc = OracleTrainer(cfg, state, batch_size=B)
  gen = SyntheticBatches(cfg.data_config, feature_configs, batch_size=B, seed=data_seed)
  out = {'config': config, 'batch_size': B, 'seed': seed, 'data_seed': data_seed, 'steps': []}
  for _ in range(2):
    losses = orc.train_step(gen.next_batch())
    step = {'losses': {k: float(v) for k, v in losses.items()}}
    for k, v in orc.last_pred.items():
      if k.startswith('logits'):
        step[k + '[:8]'] = [float(x) for x in np.asarray(v).reshape(-1)[:8]]
    out['steps'].append(step)
  return out


def main():
  import logging
  import torch
  logging.disable(logging.WARNING)
  relu, margins = torch.relu, []

  def watched_relu(x):
    a = x.detach().abs().reshape(-1, x.shape[-1])
    top = a.max(dim=0).values
    # columns that are (nearly) constant over the batch normalise to ~0: left out; the others RELATIVE to the column's
    # largest entry (the experts of MMoE run BatchNorm on the moving statistics: nothing brings their inputs to O(1))
    live = a[:, top > 1e-3] / top[top > 1e-3]
    nz = live[live > 0]
    if nz.numel():
      margins.append(float(nz.min()))
    return relu(x)

  res = []
  for c in CASES:
    del margins[:]
    torch.relu = watched_relu
    try:
      res.append(run_case(*c))
    finally:
      torch.relu = relu
    assert not margins or min(margins) > 5e-7, (c, min(margins), 'near-tie at a ReLU: pick another seed')
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'model_vectors.json')
  with open(path, 'w') as f:
    json.dump({'generator': 'tests/golden/make_model_vectors.py', 'cases': res}, f, indent=1)
  print('wrote', path)


if __name__ == '__main__':
  main()
