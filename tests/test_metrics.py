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


def _separated_auc_like_the_reference(labels, preds, keys, reduction):
  """The reference's `_separated_auc_impl` (core/metrics.py:59-108) restated: python dictionaries keyed by the
  group, sklearn.metrics.roc_auc_score per group with both classes (sklearn IS what the reference calls)."""
  from collections import defaultdict
  from sklearn import metrics as sk
  lab, pr, w = defaultdict(list), defaultdict(list), defaultdict(int)
  for y, p, k in zip(labels, preds, keys):
    lab[k].append(y)
    pr[k].append(p)
    w[k] = 1 if reduction == 'mean' else w[k] + (1 if reduction == 'mean_by_sample_num' else y)
  ms, ws = [], []
  for k in lab:
    y = np.asarray(lab[k])
    if np.all(y == 1) or np.all(y == 0):
      continue
    ms.append(sk.roc_auc_score(y, np.asarray(pr[k])))
    ws.append(w[k])
  return np.float32(np.average(ms, weights=ws)) if ms else np.float32(0.0)


@pytest.mark.parametrize('reduction', ['mean', 'mean_by_sample_num', 'mean_by_positive_num'])
@pytest.mark.parametrize('string_keys', [False, True])
def test_grouped_auc_equals_the_reference_algorithm(reduction, string_keys):
  from easyrec_amd.core.metrics import SeparatedAUC, roc_auc
  from sklearn import metrics as sk
  rng = np.random.default_rng(7)
  n = 3000
  users = rng.integers(0, 120, size=n)  # some users end up with one class only, some with a single row
  users[:40] = 1000 + np.arange(40)
  y = (rng.random(n) < 0.3).astype(np.int64)
  p = np.round(np.clip(rng.normal(0.4 + 0.15 * y, 0.2), 0, 1), 2).astype(np.float32)  # rounded: ties inside groups
  keys = np.array([b'u%d' % u for u in users], dtype=object) if string_keys else users
  m = SeparatedAUC(reduction)
  for lo in range(0, n, 700):  # streaming
    m.update(torch.from_numpy(y[lo:lo + 700]), torch.from_numpy(p[lo:lo + 700]), keys[lo:lo + 700])
  exp = _separated_auc_like_the_reference(y, p, [k for k in keys], reduction)
  assert abs(m.result() - float(exp)) < 1e-6, (m.result(), exp)
  assert abs(roc_auc(y, p) - sk.roc_auc_score(y, p)) < 1e-12
  empty = SeparatedAUC(reduction)
  empty.update(np.ones(5), np.linspace(0, 1, 5), np.arange(5))  # no key has both classes
  assert empty.result() == 0.0


def test_max_f1_follows_the_restatement():
  """core/metrics.py:25-56: tf.metrics.precision / recall of (prediction > threshold) at 200 thresholds."""
  from easyrec_amd.core.metrics import MaxF1, auc_thresholds
  rng = np.random.default_rng(3)
  n = 4000
  y = (rng.random(n) < 0.2).astype(np.float32)
  p = np.clip(rng.normal(0.3 + 0.3 * y, 0.15), 0, 1).astype(np.float32)
  m = MaxF1()
  for lo in range(0, n, 1500):
    m.update(y[lo:lo + 1500], torch.from_numpy(p[lo:lo + 1500]))
  best = 0.0
  for t in auc_thresholds(200):
    pred = p > t
    tp, fp, fn = float((pred & (y > 0)).sum()), float((pred & (y == 0)).sum()), float((~pred & (y > 0)).sum())
    prec = tp / (tp + fp) if tp + fp > 0 else 0.0
    rec = tp / (tp + fn) if tp + fn > 0 else 0.0
    best = max(best, 2 * prec * rec / (prec + rec + 1e-12))
  assert abs(m.result() - best) < 1e-9 and 0.3 < best < 1.0


def test_multi_task_metrics_are_each_tower_s_own(ref_backend):
  """A multi-task model evaluates every tower with the tower's OWN metrics_set (multi_task_model.py:143-158), not with
  eval_config.metrics_set: a tower without one reports nothing, a tower may ask for max_f1 beside auc."""
  from google.protobuf import text_format
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'mmoe_taobao_small.config'))
  towers = cfg.model_config.mmoe.task_towers
  towers[1].ClearField('metrics_set')
  text_format.Merge('metrics_set { max_f1 {} }', towers[0])
  est = EasyRecEstimator(cfg, device='cpu', batch_size=32, seed=1).build()
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=32, seed=5)
  batches = [gen.next_batch() for _ in range(2)]
  est.train_step(batches[0])
  out = est.evaluate(batches)
  assert sorted(out) == ['auc_%s' % towers[0].tower_name, 'max_f1_%s' % towers[0].tower_name], out


def test_evaluate_with_grouped_metrics(ref_backend):
  _evaluate_with_grouped_metrics('cpu')


@pytest.mark.gpu
def test_evaluate_with_grouped_metrics_on_the_gpu():
  _evaluate_with_grouped_metrics('cuda:0')


def _evaluate_with_grouped_metrics(device):
  """metrics_set { gauc } / { session_auc } / { max_f1 } through EasyRecEstimator.evaluate(): the key column is the
  RAW value of the named feature (its strings for a hashed id feature), as `feature_dict[uid_field]` in the
  reference; the result must equal the metric recomputed from the predictions by the reference's algorithm."""
  from google.protobuf import text_format
  from easyrec_amd.input.features import host_key_column
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
  text_format.Merge('metrics_set { gauc { uid_field: "C1" reduction: "mean_by_sample_num" } } '
                    'metrics_set { session_auc { session_id_field: "C2" } } metrics_set { max_f1 {} }', cfg.eval_config)
  B = 64
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  est = EasyRecEstimator(cfg, device=device, batch_size=B, seed=1).build()
  gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=B, seed=5)  # id features as raw strings
  batches = [gen.next_batch() for _ in range(4)]
  est.train_step(batches[0])
  out = est.evaluate(batches)
  assert sorted(out) == ['auc', 'gauc', 'max_f1', 'session_auc']
  # recompute from the predictions
  labels, probs, k1, k2, logits = [], [], [], [], []
  est.model._is_training, est.ctx.is_training = False, False
  for b in batches:
    pred = est.predict(b)
    probs.append(pred['probs'].detach().cpu().numpy().reshape(-1).copy())
    logits.append(pred['logits'].detach().cpu().numpy().reshape(-1).copy())
    labels.append(est.features.label(est.model._label_name).cpu().numpy().reshape(-1).copy())
    k1.append(host_key_column(est.features.schema, b, 'C1'))
    k2.append(host_key_column(est.features.schema, b, 'C2'))
  est.model._is_training, est.ctx.is_training = True, True
  labels, probs = np.concatenate(labels).astype(np.int64), np.concatenate(probs)
  k1, k2 = np.concatenate(k1), np.concatenate(k2)
  assert k1.dtype == object and len(set(k1.tolist())) > 3  # raw strings, several users
  assert abs(out['gauc'] - float(_separated_auc_like_the_reference(labels, probs, k1.tolist(), 'mean_by_sample_num'))) < 1e-6
  assert abs(out['session_auc'] - float(_separated_auc_like_the_reference(labels, probs, k2.tolist(), 'mean'))) < 1e-6
  # max_f1 reads the LOGITS (rank_model.py:424-427), not the probabilities
  from easyrec_amd.core.metrics import MaxF1
  on_logits, on_probs = MaxF1(), MaxF1()
  on_logits.update(labels, np.concatenate(logits))
  on_probs.update(labels, probs)
  assert abs(out['max_f1'] - on_logits.result()) < 1e-9 and 0.0 <= out['max_f1'] <= 1.0
  assert on_logits.result() != on_probs.result()


def test_host_key_column_variants():
  """The grouping keys of gAUC / session AUC come from the HOST batch: raw strings of a hashed id feature, the bucket
  when the batch was hashed on the host, the integer of an identity feature; a packed (device) batch has none."""
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.input.features import FeatureSchema, host_key_column
  from easyrec_amd.utils import config_util
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
  feats = list(cfg.feature_config.features)
  B = 16
  gen = SyntheticCriteo(cfg.data_config, feats, batch_size=B, seed=1)
  b = gen.next_batch()
  sch = FeatureSchema(cfg.data_config, feats, batch_size=B)
  k = host_key_column(sch, b, 'C5')
  assert k.dtype == object and len(k) == B
  j = sch.hash_single['C5']['col']
  off = np.asarray(b['str_offsets'])
  raw = np.asarray(b['str_bytes'], dtype=np.uint8).tobytes()
  assert k[3] == raw[off[j * B + 3]:off[j * B + 4]]
  hashed = {'hash_ids': np.arange(len(sch.hash_single) * B, dtype=np.int64).reshape(len(sch.hash_single), B)}
  assert np.array_equal(host_key_column(sch, hashed, 'C5'), hashed['hash_ids'][j])
  with pytest.raises(KeyError):
    host_key_column(sch, {'packed': None}, 'C5')
  with pytest.raises(KeyError):
    host_key_column(sch, b, 'F1')  # a raw feature has no key column
