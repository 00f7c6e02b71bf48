"""Dense-variable store: TF-style `get_variable` by name over ONE flat HBM buffer.

The reference builds its graph once with `tf.get_variable` / `tf.layers.dense(name=...)`
(layers/dnn.py:57-62, model/dcn.py:37-42, ...), and the optimizer then issues one ApplyAdam per
variable (compat/optimizers.py:412-416).  Here every dense variable is a view into one flat fp32
buffer (params / grads / Adam m / Adam v / per-element L2 coefficient), so the optimizer step,
gradient zeroing and the data-parallel all-reduce are each ONE launch / ONE collective.

Variables are created lazily on first use (the "build pass"), then `pack()` moves them into the
flat buffers; names follow the reference's variable scopes so checkpoints can be interchanged.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch


def _seed_for(name, base_seed):
  return (zlib.crc32(name.encode('utf-8')) ^ (base_seed * 2654435761)) & 0x7FFFFFFF


def glorot_uniform(shape, rng):
  """tf.glorot_uniform_initializer (default of tf.layers.dense / tf.get_variable)."""
  if len(shape) == 1:
    fan_in = fan_out = shape[0]
  else:
    fan_in, fan_out = shape[-2], shape[-1]
  limit = math.sqrt(6.0 / (fan_in + fan_out))
  return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def he_uniform(shape, rng):
  fan_in = shape[0] if len(shape) == 1 else shape[-2]
  limit = math.sqrt(6.0 / fan_in)
  return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def truncated_normal(shape, rng, mean=0.0, stddev=1.0):
  """tf.truncated_normal_initializer: samples beyond 2 sigma are redrawn."""
  out = rng.normal(0.0, 1.0, size=shape)
  bad = np.abs(out) > 2.0
  while bad.any():
    out[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
    bad = np.abs(out) > 2.0
  return (out * stddev + mean).astype(np.float32)


def he_normal(shape, rng):
  """tf.initializers.he_normal = VarianceScaling(2.0, 'fan_in', 'truncated_normal'): stddev sqrt(2 / fan_in) with the
  correction 0.8796 for the truncation at 2 sigma; fan_in of an [.., a, b] kernel = a * prod(leading dims)."""
  if len(shape) == 1:
    fan_in = shape[0]
  else:
    fan_in = shape[-2] * int(np.prod(shape[:-2])) if len(shape) > 2 else shape[-2]
  return truncated_normal(shape, rng, 0.0, math.sqrt(2.0 / fan_in) / 0.87962566103423978)


def zeros(shape, rng=None):
  return np.zeros(shape, dtype=np.float32)


def ones(shape, rng=None):
  return np.ones(shape, dtype=np.float32)


INITIALIZERS = {
    'glorot_uniform': glorot_uniform,
    'he_uniform': he_uniform,
    'he_normal': he_normal,
    'zeros': zeros,
    'ones': ones,
    'truncated_normal': lambda shape, rng: truncated_normal(shape, rng, 0.0, 0.05),
}


class VarStore(object):
  """name -> tensor; trainable variables share one flat buffer after `pack()`."""

  def __init__(self, device, seed=0):
    self.device = torch.device(device)
    self.seed = seed
    self._vars = OrderedDict()  # name -> dict(tensor, trainable, l2)
    self._scope = []
    self.packed = False
    self.flat = None
    self.flat_grad = None
    self.l2coef = None
    self.slots = {}  # optimizer slots over the flat buffer
    self._offsets = {}

  # -- TF-like name scopes
  def scope(self, name):
    store = self

    class _Scope(object):

      def __enter__(self_inner):
        store._scope.append(name)

      def __exit__(self_inner, *a):
        store._scope.pop()

    return _Scope()

  def full_name(self, name):
    return '/'.join([s for s in self._scope if s] + [name])

  def get_variable(self, name, shape, initializer='glorot_uniform', l2=0.0, trainable=True):
    """Create on first use (build pass), fetch afterwards."""
    full = self.full_name(name)
    rec = self._vars.get(full)
    if rec is not None:
      assert tuple(rec['tensor'].shape) == tuple(shape), \
          'variable %s: shape %s requested, %s exists' % (full, shape, tuple(rec['tensor'].shape))
      return rec['tensor']
    assert not self.packed, 'variable %s requested after pack(); run the build pass first' % full
    rng = np.random.RandomState(_seed_for(full, self.seed))
    if callable(initializer):
      value = initializer(tuple(shape), rng)
    else:
      value = INITIALIZERS[initializer](tuple(shape), rng)
    t = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32)).to(self.device)
    if trainable:
      t.requires_grad_(True)
    self._vars[full] = {'tensor': t, 'trainable': trainable, 'l2': float(l2)}
    return t

  def has(self, full_name):
    return full_name in self._vars

  def names(self):
    return list(self._vars.keys())

  def trainable_names(self):
    return [n for n, r in self._vars.items() if r['trainable']]

  def pack(self, extra_grad_floats=0):
    """Move every trainable variable into one flat buffer (+ grads + L2 coefficients).  extra_grad_floats: room
    behind the gradients (`grad_tail`) for gradients that are zeroed and all-reduced with them (`flat_grad_all`)."""
    assert not self.packed
    names = self.trainable_names()
    total, off = 0, {}
    for n in names:
      numel = self._vars[n]['tensor'].numel()
      off[n] = (total, numel)
      total += (numel + 3) // 4 * 4  # keep every view 16-byte aligned
    total = max(total, 4)
    self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
    self.flat_grad_all = torch.zeros(total + int(extra_grad_floats), dtype=torch.float32, device=self.device)
    self.flat_grad = self.flat_grad_all[:total]
    self.grad_tail = self.flat_grad_all[total:]
    self.l2coef = torch.zeros(total, dtype=torch.float32, device=self.device)
    for n in names:
      rec = self._vars[n]
      o, numel = off[n]
      t = rec['tensor']
      view = self.flat[o:o + numel].view(t.shape)
      view.copy_(t.detach())
      t.data = view
      t.grad = self.flat_grad[o:o + numel].view(t.shape)
      if rec['l2'] != 0.0:
        self.l2coef[o:o + numel] = rec['l2']
    self._offsets = off
    self.packed = True
    self.any_l2 = any(self._vars[n]['l2'] != 0.0 for n in names)
    # per-256-weight sums of 0.5 * l2 * w^2: the kernel-L2 term of the loss without a pass over the weights - the dense
    # optimizer leaves them behind for the next step (er_dense_opt_step_l2); refreshed here and after every load
    self.l2_partials = torch.zeros((total + 255) // 256, dtype=torch.float32, device=self.device) if self.any_l2 else None
    self.refresh_l2_partials()

  def refresh_l2_partials(self):
    # (called whenever the weights were written from outside the optimizer: pack, load_state_dict)
    self.version = getattr(self, 'version', 0) + 1  # kernels.Bf16Shadows re-casts the weights' bf16 shadows
    if getattr(self, 'l2_partials', None) is not None:
      from easyrec_amd import kernels
      kernels.hip().l2_partials(self.flat, self.l2coef, self.l2_partials)

  def slot(self, name):
    if name not in self.slots:
      self.slots[name] = torch.zeros_like(self.flat)
    return self.slots[name]

  def zero_grad(self):
    self.flat_grad_all.zero_()

  def check_grad_views(self):
    """Autograd must have accumulated in place into the flat gradient buffer."""
    for n in self.trainable_names():
      o, numel = self._offsets[n]
      t = self._vars[n]['tensor']
      assert t.grad is not None and t.grad.data_ptr() == self.flat_grad.data_ptr() + 4 * o, \
          'gradient of %s left the flat buffer' % n

  # -- host exchange (parity tests, checkpoints)
  def state_dict(self):
    return OrderedDict((n, r['tensor'].detach().cpu().numpy().copy()) for n, r in self._vars.items())

  def load_state_dict(self, state, strict=True):
    for n, r in self._vars.items():
      if n in state:
        with torch.no_grad():
          r['tensor'].copy_(torch.from_numpy(np.asarray(state[n], dtype=np.float32)).to(self.device))
      elif strict:
        raise KeyError('missing variable %s' % n)
    if self.packed:
      self.refresh_l2_partials()

  def grad_dict(self):
    out = OrderedDict()
    for n in self.trainable_names():
      g = self._vars[n]['tensor'].grad
      out[n] = None if g is None else g.detach().cpu().numpy().copy()
    return out

  def l2_of(self, name):
    return self._vars[name]['l2']
