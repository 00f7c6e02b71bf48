"""Host logic of round 3's grouped / multi launches on the oracle's stand-in backend (no GPU): each regrouping must give
what the one-by-one form gives, variable by variable, because only the launch count is supposed to change.
  * the towers' losses in one call (loss_builder.build_many) against build() per tower;
  * the towers' output projections (dnn.dense_parallel) and the plain towers of MultiTowerDIN / the stacks of MMoE
    (dnn.run_parallel) against the sequential calls: every variable and slot after two steps;
  * pack() leaves the batch's longest sequence so that load() never asks a device tensor for it."""
import os

import numpy as np
import pytest
import torch

from easyrec_amd.utils import config_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _states(config, B, switch, steps=2, seed=5):
  """final state_dict (with slots) and losses with the backend attribute `switch` on, then off"""
  from easyrec_amd import kernels
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
  be = kernels.hip()
  out = []
  prev = getattr(be, switch)
  try:
    for on in (True, False):
      setattr(be, switch, on)
      est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=seed).build()
      gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=seed + 8)
      for _ in range(steps):
        est.train_step(gen.next_batch())
      out.append((est.state_dict(slots=True), est.loss_values()))
  finally:
    setattr(be, switch, prev)
  return out


@pytest.mark.parametrize('config', ['mmoe_taobao_small.config', 'din_taobao_small.config', 'mmoe_backbone_taobao_small.config'])
def test_grouped_stacks_equal_the_sequential_form(ref_backend, config):
  (sa, la), (sb, lb) = _states(config, 64, 'grouped_stacks')
  for k in lb:
    assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
  assert set(sa) == set(sb)
  for k in sb:
    a, b = np.asarray(sa[k]), np.asarray(sb[k])
    assert np.allclose(a, b, rtol=2e-5, atol=2e-6), (k, float(np.max(np.abs(a - b))))


def test_build_many_equals_build_per_head(ref_backend):
  from easyrec_amd.builders import loss_builder
  from easyrec_amd.protos.loss_pb2 import LossType
  g = torch.Generator().manual_seed(3)
  specs = []
  for i in range(5):
    B = 64
    pred = torch.randn(B, generator=g) * 2
    label = (torch.rand(B, generator=g) > 0.6).float()
    w = None if i % 2 == 0 else torch.rand(B, generator=g) * (torch.rand(B, generator=g) > 0.3).float()
    loss_type = LossType.L2_LOSS if i == 3 else LossType.CLASSIFICATION
    specs.append(dict(loss_type=loss_type, label=label, pred=pred, loss_weight=w if w is not None else 0.5 + i,
                      num_class=1, loss_scale=1.0 + 0.1 * i))
  many = loss_builder.build_many(specs)
  for sp, (loss, d) in zip(specs, many):
    l1, d1 = loss_builder.build(sp['loss_type'], sp['label'], sp['pred'], sp['loss_weight'], 1, loss_scale=sp['loss_scale'])
    assert torch.equal(loss.reshape(-1), l1.reshape(-1)) and torch.equal(d, d1)


def test_pack_leaves_the_longest_sequence_of_the_batch(ref_backend):
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'din_taobao_small.config'))
  est = EasyRecEstimator(cfg, device='cpu', batch_size=32, seed=2).build()
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=32, seed=11)
  b = gen.next_batch()
  packed = est.features.pack(b)
  names = list(est.features.schema.seqs)
  assert names
  for n in names:
    assert packed['seq/%s/max' % n] == int(np.asarray(b['seq/%s/len' % n]).max())
  # load() takes it from the batch: a wrong value in the dict is what ends up in the model's padded length
  packed['seq/%s/max' % names[0]] = 1
  est.features.load(packed)
  assert est.features.seq_batch_max[names[0]] == 1
