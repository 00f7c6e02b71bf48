"""Evaluation metrics on the hot path's predictions.

AUC = `tf.metrics.auc(labels, predictions, num_thresholds=200)` as called by `RankModel.build_metric_graph`
(reference easy_rec/python/model/rank_model.py:358-373; `eval_config.metrics_set { auc {} }`).  TensorFlow's
published algorithm (tf.metrics.auc, curve='ROC', summation_method='trapezoidal'; TF is third-party and absent
from /root/reference): thresholds `[-1e-7] + [(i+1)/(T-1) for i < T-2] + [1+1e-7]` in float32; per threshold the
running tp / fn / tn / fp of `prediction > threshold`; `recall = (tp + 1e-6) / (tp + fn + 1e-6)`,
`fpr = fp / (fp + tn + 1e-6)`; `auc = sum((fpr[:-1] - fpr[1:]) * (recall[:-1] + recall[1:]) / 2)`.
The per-batch update is one HIP launch (`er_auc_update`: a histogram over "number of thresholds below the
prediction", integer atomics, so the counts are exact and order-independent); the sums are finished on the host.
"""
import numpy as np
import torch

from easyrec_amd import kernels

K_EPSILON = 1e-7


def auc_thresholds(num_thresholds):
  t = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
  return np.array([0.0 - K_EPSILON] + t + [1.0 + K_EPSILON], dtype=np.float32)


def auc_from_counts(counts):
  """counts int64 [2, T + 1] (er_auc_update) -> float (TF's float32 arithmetic on the running totals)."""
  counts = np.asarray(counts, dtype=np.int64)
  T = counts.shape[1] - 1
  # prediction > thresholds[t]  <=>  bucket > t
  suffix = np.cumsum(counts[:, ::-1], axis=1)[:, ::-1]
  tp, fp = suffix[1, 1:T + 1].astype(np.float32), suffix[0, 1:T + 1].astype(np.float32)
  fn = (counts[1].sum() - suffix[1, 1:T + 1]).astype(np.float32)
  tn = (counts[0].sum() - suffix[0, 1:T + 1]).astype(np.float32)
  eps = np.float32(1e-6)
  rec = (tp + eps) / (tp + fn + eps)
  fpr = fp / (fp + tn + eps)
  return float(np.sum((fpr[:T - 1] - fpr[1:]) * (rec[:T - 1] + rec[1:]) / np.float32(2.0), dtype=np.float32))


class AUC(object):
  """Streaming AUC with device-resident counts."""

  def __init__(self, num_thresholds=200, device='cuda:0'):
    assert num_thresholds >= 2
    self.num_thresholds = int(num_thresholds)
    self.thresholds = torch.from_numpy(auc_thresholds(self.num_thresholds)).to(device)
    self.counts = torch.zeros(2, self.num_thresholds + 1, dtype=torch.int64, device=device)

  def reset(self):
    self.counts.zero_()

  def update(self, labels, probs, weights=None):
    kernels.hip().auc_update(probs, labels, weights, self.thresholds, self.counts)

  def result(self):
    return auc_from_counts(self.counts.cpu().numpy())
