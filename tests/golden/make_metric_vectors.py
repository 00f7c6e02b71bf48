#!/usr/bin/env python
"""Grouped AUC and max-F1 from the REFERENCE'S OWN code (easy_rec/python/core/metrics.py: gauc / session_auc ->
_separated_auc_impl :59-108, max_f1 :25-56), run in the build container where /root/reference exists.

gauc's arithmetic is Python (dictionaries of per-key lists, sklearn.metrics.roc_auc_score - sklearn is installed - and
np.average) inside tf.py_func: the stand-in's py_func calls the function.  max_f1 is a threshold grid and an F1 formula
over tf's streaming precision / recall, which the stand-in accumulates as tp / (tp + fp) and tp / (tp + fn) (0 when the
denominator is 0: tf.metrics.precision / recall's documented div-no-nan).  Streams of several batches are fed; inputs and
results go to tests/golden/metric_vectors.npz; tests/test_metric_pins.py holds easyrec_amd/core/metrics.py to them.

usage: python tests/golden/make_metric_vectors.py [/root/reference]
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


class Streaming(object):
  """tf.metrics.precision / recall: value read after the updates"""

  def __init__(self, kind, labels, pred):
    self.kind, self.tp, self.other = kind, 0.0, 0.0
    self.labels, self.pred = labels, pred

  def update(self):
    lab, pred = np.asarray(self.labels.value).astype(bool), np.asarray(self.pred.value).astype(bool)
    self.tp += float((lab & pred).sum())
    self.other += float((~lab & pred).sum()) if self.kind == 'precision' else float((lab & ~pred).sum())

  def value(self):
    d = self.tp + self.other
    return self.tp / d if d > 0 else 0.0


class Lazy(object):
  """a placeholder-like tensor: arithmetic builds closures evaluated at read-out"""

  def __init__(self, fn):
    self.fn = fn

  @property
  def value(self):
    return self.fn()

  def _bin(self, other, op):
    return Lazy(lambda: op(self.value, other.value if isinstance(other, Lazy) else other))

  def __gt__(self, o): return self._bin(o, lambda a, b: a > b)
  def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
  __rmul__ = __mul__
  def __add__(self, o): return self._bin(o, lambda a, b: a + b)
  def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)


def main():
  tf = types.ModuleType('tensorflow')
  tf.__version__ = '1.15.0'
  tf.float32 = np.float32
  tf.py_func = lambda fn, inputs, out_types: (lambda: fn(*[i.value if isinstance(i, Lazy) else i for i in inputs]))
  tf.stack = lambda xs: Lazy(lambda: np.array([x.value for x in xs]))
  tf.math = types.SimpleNamespace(reduce_max=lambda x: Lazy(lambda: float(np.max(x.value))))
  tf.group = lambda ops: (lambda: [op() for op in ops])
  sys.modules['tensorflow'] = tf
  for name in ('tensorflow.python', 'tensorflow.python.ops', 'tensorflow.python.ops.array_ops', 'tensorflow.python.ops.math_ops',
               'tensorflow.python.ops.state_ops', 'tensorflow.python.ops.variable_scope', 'easy_rec', 'easy_rec.python',
               'easy_rec.python.utils', 'easy_rec.python.utils.estimator_utils', 'easy_rec.python.utils.io_util',
               'easy_rec.python.utils.shape_utils', 'easy_rec.python.core', 'easy_rec.python.core.easyrec_metrics'):
    sys.modules[name] = types.ModuleType(name)
  for sub in ('array_ops', 'math_ops', 'state_ops', 'variable_scope'):
    setattr(sys.modules['tensorflow.python.ops'], sub, sys.modules['tensorflow.python.ops.' + sub])
  sys.modules['easy_rec.python.utils.estimator_utils'].get_task_index_and_num = None
  sys.modules['easy_rec.python.utils.io_util'].read_data_from_json_path = None
  sys.modules['easy_rec.python.utils.io_util'].save_data_to_json_path = None
  sys.modules['easy_rec.python.utils.shape_utils'].get_shape_list = None
  made = []

  def streaming(kind):
    def make(labels, predictions, name=None):
      m = Streaming(kind, labels, predictions)
      made.append(m)
      return Lazy(m.value), m.update
    return make

  metrics_tf = types.SimpleNamespace(precision=streaming('precision'), recall=streaming('recall'))
  sys.modules['easy_rec.python.core.easyrec_metrics'].metrics_tf = metrics_tf
  spec = importlib.util.spec_from_file_location('ref_core_metrics', os.path.join(REF, 'easy_rec/python/core/metrics.py'))
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)

  rng = np.random.default_rng(20240927)
  out = {}
  n_batches, B = 4, 60
  keys = [rng.integers(0, 9, B) for _ in range(n_batches)]  # 9 users; some end up with a single class
  labels = [(rng.random(B) < 0.35).astype(np.float32) for _ in range(n_batches)]
  labels[0][keys[0] == 7] = 1.0
  for b in range(1, n_batches):
    labels[b][keys[b] == 7] = 1.0  # user 7: positives only (skipped)
    labels[b][keys[b] == 8] = 0.0
  labels[0][keys[0] == 8] = 0.0    # user 8: negatives only (skipped)
  preds = [np.round(rng.random(B), 2).astype(np.float32) for _ in range(n_batches)]  # (two decimals: ties occur)
  for b in range(n_batches):
    out['keys_%d' % b], out['labels_%d' % b], out['preds_%d' % b] = keys[b], labels[b], preds[b]
  for reduction in ('mean', 'mean_by_sample_num', 'mean_by_positive_num'):
    cur = {}
    lab_t, pred_t, key_t = (Lazy(lambda k=k: cur[k]) for k in ('labels', 'preds', 'keys'))
    value_op, update_op = ref.gauc(lab_t, pred_t, key_t, reduction=reduction)
    for b in range(n_batches):
      cur.update(labels=labels[b], preds=preds[b], keys=keys[b])
      update_op()
      out['gauc_%s_after_%d' % (reduction, b)] = np.float32(value_op())
  # string keys (session ids), one batch
  skeys = np.array(['s%d' % k for k in keys[0]])
  out['session_keys'] = skeys
  cur = {}
  value_op, update_op = ref.session_auc(Lazy(lambda: cur['labels']), Lazy(lambda: cur['preds']), Lazy(lambda: cur['keys']))
  cur.update(labels=labels[0], preds=preds[0], keys=skeys)
  update_op()
  out['session_auc'] = np.float32(value_op())
  # max_f1 over the same stream
  cur = {}
  f1, f1_update = ref.max_f1(Lazy(lambda: cur['labels']), Lazy(lambda: cur['preds']))
  for b in range(n_batches):
    cur.update(labels=labels[b], preds=preds[b])
    f1_update()
    out['max_f1_after_%d' % b] = np.float32(f1.value)
  path = os.path.join(HERE, 'metric_vectors.npz')
  np.savez(path, **out)
  print('wrote %s: %d arrays; gauc mean %.6f, max_f1 %.6f' % (path, len(out), out['gauc_mean_after_3'], out['max_f1_after_3']))


if __name__ == '__main__':
  main()
