"""Evaluation metrics on the hot path's predictions.

Also: gAUC / session AUC (`SeparatedAUC` host-side as in the reference, `DeviceSeparatedAUC` the same numbers from device
kernels: what evaluate() uses) and max F1 (`MaxF1` on the host, `DeviceMaxF1` from the AUC histogram kernel: what
evaluate() uses).

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


# ------------------------------------------------------------------------------------------------------------------
# Grouped AUCs (reference easy_rec/python/core/metrics.py:59-108 `_separated_auc_impl`, :260-297 `gauc` /
# `session_auc`; wired by model/rank_model.py:376-420).  The reference accumulates (label, prediction, key) in Python
# dictionaries inside a tf.py_func and, at read-out, averages `sklearn.metrics.roc_auc_score` over the keys that have
# both classes - host work there, host work here: the rows are kept as arrays and grouped once with a sort.
# ------------------------------------------------------------------------------------------------------------------
def roc_auc(labels, predictions):
  """Area under the ROC curve of binary `labels` = the Mann-Whitney statistic with tied predictions counted half
  (what sklearn.metrics.roc_auc_score returns for a binary target).  Needs both classes."""
  labels = np.asarray(labels).reshape(-1) != 0
  predictions = np.asarray(predictions, dtype=np.float64).reshape(-1)
  n_pos = int(labels.sum())
  n_neg = labels.size - n_pos
  assert n_pos > 0 and n_neg > 0, 'roc_auc: only one class present'
  order = np.argsort(predictions, kind='mergesort')
  p = predictions[order]
  # average rank (1-based) of every run of equal predictions
  starts = np.flatnonzero(np.concatenate([[True], p[1:] != p[:-1]]))
  ends = np.concatenate([starts[1:], [p.size]])
  run_rank = (starts + 1 + ends) / 2.0
  ranks = np.repeat(run_rank, ends - starts)
  pos_rank_sum = ranks[labels[order]].sum()
  return float((pos_rank_sum - n_pos * (n_pos + 1) / 2.0) / (n_pos * float(n_neg)))


class SeparatedAUC(object):
  """AUC per key (user, session), reduced by 'mean' | 'mean_by_sample_num' | 'mean_by_positive_num'."""

  REDUCTIONS = ('mean', 'mean_by_sample_num', 'mean_by_positive_num')

  def __init__(self, reduction='mean'):
    assert reduction in self.REDUCTIONS, 'reduction method must in mean | mean_by_sample_num | mean_by_positive_num'
    self.reduction = reduction
    self.reset()

  def reset(self):
    self._labels, self._preds, self._keys = [], [], []

  def update(self, labels, predictions, keys):
    def host(x):
      return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    labels, predictions, keys = host(labels).reshape(-1), host(predictions).reshape(-1), host(keys).reshape(-1)
    assert labels.size == predictions.size == keys.size
    self._labels.append(labels.astype(np.int64))
    self._preds.append(predictions.astype(np.float64))
    self._keys.append(keys)

  def result(self):
    if not self._labels:
      return 0.0
    labels, preds = np.concatenate(self._labels), np.concatenate(self._preds)
    keys = np.concatenate([k.astype(object) for k in self._keys]) if any(k.dtype == object for k in self._keys) \
        else np.concatenate(self._keys)
    _, inv = np.unique(keys, return_inverse=True)
    order = np.argsort(inv, kind='mergesort')
    bounds = np.flatnonzero(np.concatenate([[True], inv[order][1:] != inv[order][:-1], [True]]))
    metrics, weights = [], []
    for b, e in zip(bounds[:-1], bounds[1:]):
      idx = order[b:e]
      lab = labels[idx]
      n_pos = int((lab != 0).sum())
      if n_pos == 0 or n_pos == lab.size:  # (metrics.py:93-94: keys with one class are skipped)
        continue
      metrics.append(roc_auc(lab, preds[idx]))
      weights.append(1 if self.reduction == 'mean' else lab.size if self.reduction == 'mean_by_sample_num' else lab.sum())
    if not metrics:
      return 0.0
    return float(np.average(metrics, weights=weights).astype(np.float32))


class DeviceSeparatedAUC(object):
  """SeparatedAUC with the rows kept and reduced on the device: labels and predictions never leave it, the keys arrive
  as one int64 per example (integer ids as they are, strings as a 64-bit digest - grouping only needs equality), and
  result() is two device sorts + er_grouped_auc (kernels.py grouped_auc).  Same numbers as SeparatedAUC."""

  def __init__(self, reduction='mean', device='cpu'):
    assert reduction in SeparatedAUC.REDUCTIONS, 'reduction method must in mean | mean_by_sample_num | mean_by_positive_num'
    self.reduction, self.device = reduction, torch.device(device)
    self.reset()

  def reset(self):
    self._labels, self._preds, self._keys = [], [], []

  @staticmethod
  def key_codes(keys):
    """One int64 per key: integers unchanged, anything else (the raw bytes of a string feature) by a 64-bit digest."""
    if torch.is_tensor(keys):
      return keys.reshape(-1).to(torch.int64)
    keys = np.asarray(keys).reshape(-1)
    if keys.dtype.kind in 'iu':
      return torch.from_numpy(keys.astype(np.int64))
    import hashlib
    def digest(k):
      raw = k if isinstance(k, bytes) else str(k).encode('utf-8')
      return int.from_bytes(hashlib.blake2b(raw, digest_size=8).digest(), 'little', signed=True)
    return torch.tensor([digest(k) for k in keys.tolist()], dtype=torch.int64)

  def update(self, labels, predictions, keys):
    labels = torch.as_tensor(labels).reshape(-1).to(self.device, torch.float32)
    predictions = torch.as_tensor(predictions).detach().reshape(-1).to(self.device, torch.float32)
    codes = self.key_codes(keys).to(self.device)
    assert labels.numel() == predictions.numel() == codes.numel()
    self._labels.append(labels.clone())
    self._preds.append(predictions.clone())
    self._keys.append(codes)

  def result(self):
    if not self._labels:
      return 0.0
    total, wsum, _ = kernels.hip().grouped_auc(torch.cat(self._keys), torch.cat(self._preds), torch.cat(self._labels),
                                               SeparatedAUC.REDUCTIONS.index(self.reduction))
    return float(np.float32(total / wsum)) if wsum > 0 else 0.0


def gauc(reduction='mean'):
  return SeparatedAUC(reduction)


def session_auc(reduction='mean'):
  return SeparatedAUC(reduction)


class DeviceMaxF1(object):
  """MaxF1 with the streaming counts on the device: `prediction > threshold` against the same 200 thresholds is the
  histogram er_auc_update already keeps (counts[label][number of thresholds below the prediction]); tp / fp / fn per
  threshold are its suffix sums, finished on the host at result().  Same numbers as MaxF1."""

  def __init__(self, num_thresholds=200, device='cpu'):
    self.n = int(num_thresholds)
    self.thresholds = torch.from_numpy(auc_thresholds(self.n)).to(device)
    self.counts = torch.zeros(2, self.n + 1, dtype=torch.int64, device=device)

  def reset(self):
    self.counts.zero_()

  def update(self, labels, predictions):
    dev = self.counts.device
    labels = torch.as_tensor(labels).reshape(-1).to(dev, torch.float32)
    predictions = torch.as_tensor(predictions).detach().reshape(-1).to(dev, torch.float32)
    kernels.hip().auc_update(predictions.contiguous(), labels.contiguous(), None, self.thresholds, self.counts)

  def result(self):
    counts = self.counts.cpu().numpy().astype(np.float64)
    suffix = np.cumsum(counts[:, ::-1], axis=1)[:, ::-1]
    tp, fp = suffix[1, 1:self.n + 1], suffix[0, 1:self.n + 1]   # prediction > thresholds[t]  <=>  bucket > t
    fn = counts[1].sum() - tp
    with np.errstate(divide='ignore', invalid='ignore'):
      p = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
      r = np.where(tp + fn > 0, tp / (tp + fn), 0.0)
    return float(np.max(2 * p * r / (p + r + 1e-12)))


class MaxF1(object):
  """Largest F1 over 200 thresholds (reference core/metrics.py:25-56): streaming tp / fp / fn per threshold
  (tf.metrics.precision / recall: 0 when their denominator is 0), f1 = 2 p r / (p + r + 1e-12)."""

  def __init__(self, num_thresholds=200):
    self.thresholds = auc_thresholds(num_thresholds)  # float32, like the predictions they are compared with
    self.reset()

  def reset(self):
    n = self.thresholds.size
    self.tp, self.fp, self.fn = np.zeros(n), np.zeros(n), np.zeros(n)

  def update(self, labels, predictions):
    def host(x):
      return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    lab = host(labels).reshape(-1) != 0
    pred = host(predictions).reshape(-1).astype(np.float32)[None, :] > self.thresholds[:, None]
    self.tp += (pred & lab[None, :]).sum(axis=1)
    self.fp += (pred & ~lab[None, :]).sum(axis=1)
    self.fn += (~pred & lab[None, :]).sum(axis=1)

  def result(self):
    with np.errstate(divide='ignore', invalid='ignore'):
      p = np.where(self.tp + self.fp > 0, self.tp / (self.tp + self.fp), 0.0)
      r = np.where(self.tp + self.fn > 0, self.tp / (self.tp + self.fn), 0.0)
    return float(np.max(2 * p * r / (p + r + 1e-12)))
