#!/usr/bin/env python
"""F1-reweighted and pairwise loss values from the REFERENCE'S OWN code (easy_rec/python/loss/f1_reweight_loss.py:10-39,
loss/pairwise_loss.py:15-70 - the two losses of the reference's samples/model_config/multi_tower_on_taobao.config), run in
the build container where /root/reference exists, on a numpy (fp64) stand-in for the ~20 tf ops they call;
tf.losses.sigmoid_cross_entropy is the documented one (elementwise max(z, 0) - z y + log1p(exp(-|z|)), times the
broadcast weights, summed and divided by the number of non-zero weights).  Besides the values, the fixture holds the
central-difference gradient of THE REFERENCE'S forward with respect to every logit - the negatives' weight of the F1
loss is a function of the logits, and TensorFlow differentiates through it - which tests/test_loss_pins.py compares with
the gradient easyrec_amd/builders/loss_builder.py seeds the backward pass with.

usage: python tests/golden/make_loss_vectors.py [/root/reference]
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


class _T(np.ndarray):

  @property
  def shape(self):
    s = np.ndarray.shape.__get__(self)

    class S(tuple):
      def as_list(self):
        return list(self)
    return S(s)


def t(x):
  return np.array(x, dtype=np.float64).view(_T)  # (a copy: the reference writes `logits /= temperature`)


def make_tf():
  tf = types.ModuleType('tensorflow')
  tf.__version__ = '1.15.0'
  tf.float32 = np.float64

  def sce(labels, logits, weights=1.0, label_smoothing=0, **kw):
    assert not label_smoothing
    z, y = np.asarray(logits, dtype=np.float64), np.asarray(labels, dtype=np.float64)
    per = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    w = np.broadcast_to(np.asarray(weights, dtype=np.float64), per.shape)
    present = float((w != 0).sum())
    return float((per * w).sum() / present) if present > 0 else 0.0

  tf.losses = types.SimpleNamespace(sigmoid_cross_entropy=sce)
  tf.nn = types.SimpleNamespace(sigmoid=lambda x: t(1.0 / (1.0 + np.exp(-np.asarray(x)))))
  tf.expand_dims = lambda x, axis: t(np.expand_dims(np.asarray(x), axis))
  tf.to_float = lambda x: t(x)
  tf.shape = lambda x: np.asarray(np.asarray(x).shape)
  tf.reduce_sum = lambda x, axis=None: t(np.sum(np.asarray(x), axis=axis))
  tf.tile = lambda x, m: t(np.tile(np.asarray(x), [int(v) for v in m]))
  tf.where = lambda c, a, b: t(np.where(c, a, b))
  tf.equal = lambda a, b: np.asarray(a) == np.asarray(b)
  tf.ones_like = lambda x: t(np.ones_like(np.asarray(x, dtype=np.float64)))
  tf.cast = lambda x, d: t(x)
  tf.math = types.SimpleNamespace(subtract=lambda a, b: t(np.asarray(a) - np.asarray(b)))
  tf.greater = lambda a, b: np.asarray(a) > np.asarray(b)
  tf.logical_and = np.logical_and
  tf.boolean_mask = lambda x, m: t(np.asarray(x)[np.asarray(m)])
  tf.size = lambda x: int(np.asarray(x).size)
  tf.summary = types.SimpleNamespace(scalar=lambda *a, **k: None)
  tf.is_numeric_tensor = lambda x: isinstance(x, np.ndarray)
  tf.stack = lambda xs: np.asarray([int(v) for v in xs])
  return tf


def load(rel, name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
  m = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(m)
  return m


def numgrad(f, z, h=1e-6):
  g = np.zeros_like(z)
  for i in range(z.size):
    a, b = z.copy(), z.copy()
    a[i] += h
    b[i] -= h
    g[i] = (f(a) - f(b)) / (2 * h)
  return g


def main():
  sys.modules['tensorflow'] = make_tf()
  for name in ('tensorflow.python', 'tensorflow.python.ops', 'tensorflow.python.ops.losses', 'tensorflow.python.ops.losses.losses_impl',
               'easy_rec', 'easy_rec.python', 'easy_rec.python.loss', 'easy_rec.python.loss.focal_loss', 'easy_rec.python.utils',
               'easy_rec.python.utils.shape_utils'):
    sys.modules[name] = types.ModuleType(name)
  sys.modules['tensorflow.python.ops.losses.losses_impl'].compute_weighted_loss = None
  sys.modules['easy_rec.python.loss.focal_loss'].sigmoid_focal_loss_with_logits = None
  sys.modules['easy_rec.python.utils.shape_utils'].get_shape_list = lambda x, rank=None: list(np.asarray(x).shape)
  f1 = load('easy_rec/python/loss/f1_reweight_loss.py', 'ref_f1_reweight_loss')
  pw = load('easy_rec/python/loss/pairwise_loss.py', 'ref_pairwise_loss')
  rng = np.random.default_rng(20240928)
  B = 40
  z = rng.standard_normal(B) * 1.5
  y = (rng.random(B) < 0.3).astype(np.float64)
  w = np.round(rng.random(B) * 2, 1)
  w[rng.random(B) < 0.15] = 0.0
  out = {'logits': z, 'labels': y, 'weights': w}
  for tag, beta2, weights in (('b1', 1.0, None), ('b225', 2.25, None), ('b05_w', 0.5, w)):
    fn = lambda zz: f1.f1_reweight_sigmoid_cross_entropy(t(y), t(zz), beta2, weights=None if weights is None else t(weights))  # noqa: E731
    out['f1_%s' % tag], out['f1_%s_grad' % tag] = fn(z), numgrad(fn, z)
  for tag, margin, temp, weights in (('plain', 0, 1.0, 1.0), ('margin_temp', 0.3, 2.0, 1.0), ('weighted', 0, 1.0, w)):
    fn = lambda zz: pw.pairwise_loss(t(y), t(zz), margin=margin, temperature=temp,  # noqa: E731
                                     weights=weights if isinstance(weights, float) else t(weights))
    out['pw_%s' % tag], out['pw_%s_grad' % tag] = fn(z), numgrad(fn, z)
  path = os.path.join(HERE, 'loss_vectors.npz')
  np.savez(path, **out)
  print('wrote %s: f1 %.6f / %.6f / %.6f, pairwise %.6f / %.6f / %.6f' % (
      path, out['f1_b1'], out['f1_b225'], out['f1_b05_w'], out['pw_plain'], out['pw_margin_temp'], out['pw_weighted']))


if __name__ == '__main__':
  main()
