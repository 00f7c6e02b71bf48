"""Streaming AUC (tf.metrics.auc as used by RankModel.build_metric_graph) - SURVEY.md 8f rank 4."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _tf_auc_restated(labels, probs, num_thresholds):
  """Direct restatement of TensorFlow's published tf.metrics.auc update + value (float32 like TF's variables)."""
  from easyrec_amd.core.metrics import auc_thresholds
  t = auc_thresholds(num_thresholds)
  y = np.asarray(labels) != 0
  p = np.asarray(probs, dtype=np.float32)
  pred_pos = p[None, :] > t[:, None]
  tp = (pred_pos & y[None, :]).sum(axis=1).astype(np.float32)
  fp = (pred_pos & ~y[None, :]).sum(axis=1).astype(np.float32)
  fn = (~pred_pos & y[None, :]).sum(axis=1).astype(np.float32)
  tn = (~pred_pos & ~y[None, :]).sum(axis=1).astype(np.float32)
  eps = np.float32(1e-6)
  rec = (tp + eps) / (tp + fn + eps)
  fpr = fp / (fp + tn + eps)
  n = num_thresholds
  return float(np.sum((fpr[:n - 1] - fpr[1:]) * (rec[:n - 1] + rec[1:]) / np.float32(2.0), dtype=np.float32))


def test_auc_known_answer_from_the_tensorflow_docs(ref_backend):
  """tf.keras.metrics.AUC(num_thresholds=3) on y_true [0,0,1,1], y_pred [0,0.5,0.3,0.9] -> 0.75 (TF API docs: the
  same thresholds and trapezoid as tf.metrics.auc)."""
  from easyrec_amd.core.metrics import AUC
  m = AUC(3, 'cpu')
  m.update(torch.tensor([0., 0., 1., 1.]), torch.tensor([0., 0.5, 0.3, 0.9]))
  assert abs(m.result() - 0.75) < 1e-6
  assert abs(_tf_auc_restated([0, 0, 1, 1], [0, 0.5, 0.3, 0.9], 3) - 0.75) < 1e-6


@pytest.mark.parametrize('n,nt', [(5000, 200), (333, 17), (1, 200)])
def test_auc_counts_follow_the_restatement(ref_backend, n, nt):
  from easyrec_amd.core.metrics import AUC
  rng = np.random.default_rng(n)
  y = (rng.random(n) < 0.3).astype(np.float32)
  p = np.clip(rng.normal(0.4 + 0.2 * y, 0.2), 0, 1).astype(np.float32)
  p[:: 7] = np.float32(0.5)  # ties with a threshold value region
  m = AUC(nt, 'cpu')
  for lo in range(0, n, 1000):  # streaming over batches
    m.update(torch.from_numpy(y[lo:lo + 1000]), torch.from_numpy(p[lo:lo + 1000]))
  assert abs(m.result() - _tf_auc_restated(y, p, nt)) < 1e-6
  if n >= 5000:  # 200 thresholds approximate the exact rank statistic well on a smooth score distribution
    from scipy.stats import rankdata
    r = rankdata(p)
    exact = (r[y > 0].sum() - (y > 0).sum() * ((y > 0).sum() + 1) / 2) / ((y > 0).sum() * (y == 0).sum())
    assert abs(m.result() - exact) < 5e-3


def test_evaluate_returns_auc(ref_backend):
  """EasyRecEstimator.evaluate(): eval-mode forward (moving statistics) + streaming AUC per head."""
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  for config, keys in (('deepfm_criteo_small.config', ['auc']), ('mmoe_taobao_small.config', ['auc_ctr', 'auc_cvr'])):
    cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
    B = 32
    est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=1).build()
    gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=5)
    batches = [gen.next_batch() for _ in range(3)]
    for b in batches:
      est.train_step(b)
    before = est.state_dict()
    out = est.evaluate(batches)
    assert sorted(out) == sorted(keys)
    assert all(0.0 <= v <= 1.0 for v in out.values()), out
    after = est.state_dict()
    for k in before:  # evaluation leaves every variable (moving statistics included) untouched
      assert np.array_equal(before[k], after[k]), k
    est.train_step(batches[0])  # and training continues
