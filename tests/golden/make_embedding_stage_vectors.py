#!/usr/bin/env python
"""Golden vectors for the EMBEDDING STAGE and the sparse optimizer from the reference's own code (run in the build
container, where /root/reference exists).

TensorFlow cannot be installed here, but the reference holds its own Python for this part of the path - plain functions
over a small vocabulary of TF ops.  This script EXECUTES them, unmodified (the files are loaded from where they lie),
against a numpy stand-in for the `tensorflow.python.*` modules they import, and stores seeded inputs and their outputs:

  easy_rec/python/compat/embedding_ops.py:15-162        _prune_invalid_ids, _prune_invalid_weights,
                                                        safe_embedding_lookup_sparse  (ids < 0 pruned; weights <= 0
                                                        pruned unless combiner == 'sum'; empty rows filled with id 0 and
                                                        zeroed afterwards, or given `default_id`; rank-3 ids)
  easy_rec/python/compat/feature_column/feature_column.py:189-244   embedding_lookup_ragged  (unique -> gather ->
                                                        (x w) -> segment sums; mean = / sum w, sqrtn = / sqrt(sum w^2))
  easy_rec/python/compat/feature_column/feature_column.py:248-357   embedding_parallel_lookup on W = 1, 2, 4 simulated
                                                        Horovod ranks (threads; hvd.alltoall exchanges between them):
                                                        unique ids -> owner = id % W -> all-to-all ids -> local row =
                                                        id / W -> gather -> all-to-all rows -> stitch -> segment sums
  easy_rec/python/compat/adam_s.py:185-213, 235-246     AdamOptimizerS._apply_sparse_shared / _finish in float32 (the
                                                        `lazy_adam_optimizer` of builders/optimizer_builder.py:91-101):
                                                        three consecutive steps on one table
  easy_rec/python/compat/regularizers.py:76-108, 138-208   l2_regularizer, sum_regularizer, apply_regularization
  easy_rec/python/compat/optimizers.py:453-481          _get_grad_norm (IndexedSlices values UN-merged; embedding-parallel
                                                        partial norms all-reduced over simulated ranks) and the multiplier
                                                        of the clip at :365-376 (clip_by_global_norm(use_norm=...))

What the stand-in supplies are TensorFlow's own ops under their documented semantics (unique keeps first occurrences;
segment_sum / sparse_segment_* over sorted segment ids; sparse_fill_empty_rows; sparse_retain; sparse_reshape;
embedding_lookup_sparse's sum / mean / sqrtn; dynamic_partition / parallel_dynamic_stitch; l2_loss = sum(x^2) / 2;
clip_by_global_norm's scale = clip * min(1 / norm, 1 / clip)).  What is PINNED is everything the reference composes
from them: which ids and weights are pruned under which combiner, what an empty row becomes, the combiner formulas of
the ragged lookup, the routing arithmetic (id % W, id / W, which rank holds which row), the order of the optimizer's
float32 operations, the shape of the norm (un-merged slices, partial norms summed across ranks).

tests/test_embedding_stage_pins.py holds the oracle, the product's lookup / routing / optimizer entry points (on the
stand-in backend, and - `-m gpu` - the HIP kernels) to tests/golden/embedding_stage_vectors.npz.

usage: python tests/golden/make_embedding_stage_vectors.py [/root/reference]
"""
import contextlib
import importlib.util
import os
import sys
import threading
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------ tensors
class Dimension(int):

  @property
  def value(self):
    return int(self)

  def __sub__(self, other):
    return Dimension(int(self) - int(other))


class TensorShape(list):

  def concatenate(self, other):
    return TensorShape(list(self) + list(other))

  def as_list(self):
    return [None if d is None else int(d) for d in self]

  @property
  def ndims(self):
    return len(self)

  def __getitem__(self, i):
    r = list.__getitem__(self, i)
    return TensorShape(r) if isinstance(i, slice) else r


class Tensor(np.ndarray):
  """numpy array answering the TensorShape calls the reference makes (`x.get_shape()[0].value`, `set_shape`)"""

  def get_shape(self):
    return TensorShape(Dimension(d) for d in np.ndarray.shape.__get__(self))

  def set_shape(self, shape):
    pass


def T(x, dtype=None):
  a = np.asarray(x, dtype=dtype)
  return a.view(Tensor) if a.ndim else np.asarray([a]).view(Tensor).reshape(())


class SparseTensor(object):

  def __init__(self, indices, values, dense_shape):
    self.indices = T(indices, np.int64).reshape(-1, len(np.asarray(dense_shape).reshape(-1)))
    self.values = T(values)
    self.dense_shape = T(dense_shape, np.int64)


class RaggedTensor(object):
  """row-partitioned values: what `embedding_lookup_ragged` reads (`value_rowids()`, `flat_values`, `row_lengths()`)"""

  def __init__(self, values, row_lengths):
    self.flat_values = self.values = T(values)
    self._lens = np.asarray(row_lengths, dtype=np.int64)

  def value_rowids(self):
    return T(np.repeat(np.arange(len(self._lens)), self._lens), np.int64)

  def row_lengths(self):
    return T(self._lens)


class _Any(object):
  """whatever else the reference's modules import and touch at import time only"""

  def __init__(self, *a, **k):
    pass

  def __call__(self, *a, **k):
    if len(a) == 1 and callable(a[0]) and not k:  # used as a decorator
      return a[0]
    return _Any()

  def __getattr__(self, name):
    if name.startswith('__') and name.endswith('__'):
      raise AttributeError(name)
    return _Any()

  def __iter__(self):
    return iter(())

  def __mro_entries__(self, bases):
    return (object,)


class AnyModule(types.ModuleType):

  def __getattr__(self, name):
    if name.startswith('__') and name.endswith('__'):
      raise AttributeError(name)
    v = _Any()
    setattr(self, name, v)
    return v


class _Finder(object):
  """any not yet fabricated `tensorflow.*` / `horovod.*` / `easy_rec.*` module the reference files import"""

  @staticmethod
  def find_spec(name, path=None, target=None):
    if name.split('.')[0] in ('tensorflow', 'horovod', 'easy_rec'):
      return importlib.util.spec_from_loader(name, _Finder, is_package=True)
    return None

  @staticmethod
  def create_module(spec):
    return module(spec.name)

  @staticmethod
  def exec_module(mod):
    pass


def module(name, **attrs):
  m = sys.modules.get(name)
  if m is None:
    m = AnyModule(name)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, leaf = name.rpartition('.')
    if parent:
      setattr(module(parent), leaf, m)
  for k, v in attrs.items():
    setattr(m, k, v)
  return m


# ------------------------------------------------------------------------------------------------ the ops
def unique(x):
  """tf.unique: values in order of first occurrence, idx into them"""
  x = np.asarray(x)
  seen, vals, idx = {}, [], np.zeros(len(x), dtype=np.int32)
  for i, v in enumerate(x.tolist()):
    if v not in seen:
      seen[v] = len(vals)
      vals.append(v)
    idx[i] = seen[v]
  return T(np.asarray(vals, dtype=x.dtype)), T(idx)


def segment_sum(data, segment_ids, name=None, num_segments=None):
  data, seg = np.asarray(data), np.asarray(segment_ids).astype(np.int64)
  assert np.all(np.diff(seg) >= 0), 'segment ids must be sorted'
  n = int(num_segments) if num_segments is not None else (int(seg.max()) + 1 if len(seg) else 0)
  out = np.zeros((n,) + data.shape[1:], dtype=data.dtype)
  for i in range(len(seg)):  # in row order, like the CPU kernel
    out[seg[i]] = out[seg[i]] + data[i]
  return T(out)


def sparse_segment(kind):

  def fn(data, indices, segment_ids, name=None, num_segments=None):
    data, idx, seg = np.asarray(data), np.asarray(indices).astype(np.int64), np.asarray(segment_ids).astype(np.int64)
    s = segment_sum(data[idx], seg, num_segments=num_segments)
    if kind == 'sum':
      return s
    cnt = np.zeros(len(s), dtype=data.dtype)
    np.add.at(cnt, seg, 1)
    cnt = cnt.reshape((-1,) + (1,) * (data.ndim - 1))
    den = cnt if kind == 'mean' else np.sqrt(cnt)
    return T(np.where(cnt > 0, np.asarray(s) / np.where(cnt > 0, den, 1), 0).astype(data.dtype))

  return fn


def sparse_retain(sp, keep):
  keep = np.asarray(keep, dtype=bool)
  return SparseTensor(np.asarray(sp.indices)[keep], np.asarray(sp.values)[keep], sp.dense_shape)


def sparse_reshape(sp, shape):
  shape = [int(s) for s in shape]
  flat = np.ravel_multi_index(tuple(np.asarray(sp.indices).T), tuple(int(d) for d in sp.dense_shape)) \
      if len(sp.indices) else np.zeros(0, dtype=np.int64)
  idx = np.stack(np.unravel_index(flat, shape), axis=1) if len(flat) else np.zeros((0, len(shape)), dtype=np.int64)
  return SparseTensor(idx, sp.values, shape)


def sparse_fill_empty_rows(sp, default):
  """rows without an entry get (row, 0) = default; entries stay in row-major order; the indicator of the filled rows"""
  n_rows = int(sp.dense_shape[0])
  idx, vals = np.asarray(sp.indices), np.asarray(sp.values)
  present = np.zeros(n_rows, dtype=bool)
  present[idx[:, 0]] = True
  add = np.nonzero(~present)[0]
  all_idx = np.concatenate([idx, np.stack([add, np.zeros_like(add)], axis=1)]) if len(add) else idx
  all_vals = np.concatenate([vals, np.full(len(add), default, dtype=vals.dtype)]) if len(add) else vals
  order = np.lexsort((all_idx[:, 1], all_idx[:, 0]))
  return SparseTensor(all_idx[order], all_vals[order], sp.dense_shape), T(~present)


def embedding_lookup_sparse(params, sp_ids, sp_weights, combiner='mean', partition_strategy='mod', name=None,
                            max_norm=None):
  """tf.nn.embedding_lookup_sparse (documented): per row of sp_ids the weighted sum of the rows of `params`, / sum(w)
  for mean, / sqrt(sum(w^2)) for sqrtn; weights None = 1"""
  assert max_norm is None
  params = np.asarray(params)
  seg = np.asarray(sp_ids.indices)[:, 0]
  ids = np.asarray(sp_ids.values).astype(np.int64)
  w = np.ones(len(ids), dtype=params.dtype) if sp_weights is None else np.asarray(sp_weights.values).astype(params.dtype)
  n_rows = int(seg.max()) + 1 if len(seg) else 0
  emb = params[ids] * w[:, None]
  out = np.asarray(segment_sum(emb, seg, num_segments=n_rows))
  if combiner == 'mean':
    out = out / np.asarray(segment_sum(w, seg, num_segments=n_rows))[:, None]
  elif combiner == 'sqrtn':
    out = out / np.sqrt(np.asarray(segment_sum(w * w, seg, num_segments=n_rows)))[:, None]
  else:
    assert combiner == 'sum'
  return T(out)


def dynamic_partition(data, partitions, num_partitions):
  data, p = np.asarray(data), np.asarray(partitions)
  return [T(data[p == i]) for i in range(int(num_partitions))]


def parallel_dynamic_stitch(indices, data, name=None):
  n = sum(len(i) for i in indices)
  data = [np.asarray(d) for d in data]
  out = np.zeros((n,) + data[0].shape[1:], dtype=data[0].dtype)
  for i, d in zip(indices, data):
    out[np.asarray(i).astype(np.int64)] = d
  return T(out)


def split(value, num_or_size_splits, axis=0):
  value = np.asarray(value)
  if np.ndim(num_or_size_splits) == 0:
    return [T(p) for p in np.split(value, int(num_or_size_splits), axis=axis)]
  sizes = [int(s) for s in np.asarray(num_or_size_splits).reshape(-1)]
  return [T(p) for p in np.split(value, np.cumsum(sizes)[:-1], axis=axis)]


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
  yield (name or default_name or 'scope') + '/'


@contextlib.contextmanager
def _ctx(*a, **k):
  yield None


# ------------------------------------------------------------------------------------------------ simulated Horovod
class HvdSim(object):
  """W ranks as threads: hvd.alltoall / grouped_allreduce exchange between them at barriers"""

  Sum = 'sum'

  def __init__(self):
    self.world = 1
    self.local = threading.local()
    self.log = {}
    self.compression = types.SimpleNamespace(NoneCompressor=None)

  def start(self, world):
    self.world = world
    self.barrier = threading.Barrier(world)
    self.slots = [None] * world
    self.log = {r: [] for r in range(world)}

  def size(self):
    return self.world

  def rank(self):
    return self.local.rank

  def alltoall(self, tensor, splits=None):
    r, W = self.local.rank, self.world
    tensor = np.asarray(tensor)
    splits = np.asarray(splits).reshape(-1).astype(np.int64)
    self.slots[r] = (tensor, np.concatenate([[0], np.cumsum(splits)]))
    self.barrier.wait()
    parts = [self.slots[s][0][self.slots[s][1][r]:self.slots[s][1][r + 1]] for s in range(W)]
    recv = np.concatenate(parts) if parts else tensor[:0]
    lens = np.asarray([len(p) for p in parts], dtype=np.int64)
    self.barrier.wait()
    self.log[r].append({'sent': tensor.copy(), 'splits': splits.copy(), 'received': recv.copy()})
    return T(recv), T(lens)

  def grouped_allreduce(self, tensors, op=None, compression=None):
    r = self.local.rank
    self.slots[r] = [np.asarray(t) for t in tensors]
    self.barrier.wait()
    out = [T(sum(self.slots[s][i] for s in range(self.world))) for i in range(len(tensors))]
    self.barrier.wait()
    return out

  def run(self, world, fn):
    self.start(world)
    res, err = [None] * world, []

    def body(rank):
      self.local.rank = rank
      try:
        res[rank] = fn(rank)
      except BaseException as e:  # noqa: BLE001
        err.append(e)
        self.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
      t.start()
    for t in ts:
      t.join()
    if err:
      raise err[0]
    return res


HVD = HvdSim()


# ------------------------------------------------------------------------------------------------ the tf stand-in
class IndexedSlices(object):

  def __init__(self, values, indices, dense_shape=None):
    self.values, self.indices, self.dense_shape = T(values), T(indices), dense_shape


class Variable(object):
  """what the optimizer touches of a variable: a name, a dtype with `base_dtype`, an array"""

  def __init__(self, value, name):
    self.a = np.array(value)
    self.name = name
    self.dtype = types.SimpleNamespace(base_dtype=self.a.dtype)

  def assign(self, value, use_locking=False):
    self.a[...] = np.asarray(value, dtype=self.a.dtype)
    return self

  def __mul__(self, other):
    return T(self.a * np.asarray(other, dtype=self.a.dtype))


class Optimizer(object):
  """tf.train.Optimizer as far as AdamOptimizerS uses it"""

  def __init__(self, use_locking, name):
    self._use_locking, self._name = use_locking, name
    self._slots, self._non_slot = {}, {}

  def _call_if_callable(self, p):
    return p() if callable(p) else p

  def _create_non_slot_variable(self, initial_value, name, colocate_with):
    self._non_slot[name] = Variable(np.float32(initial_value), name)

  def _get_non_slot_variable(self, name, graph=None):
    return self._non_slot[name]

  def _zeros_slot(self, var, slot_name, op_name):
    self._slots.setdefault(slot_name, {})[var.name] = Variable(np.zeros_like(var.a), var.name + '/' + op_name +
                                                               ('' if slot_name == 'm' else '_1'))

  def get_slot(self, var, name):
    return self._slots[name][var.name]


def install():
  """the `tensorflow.python.*` modules of the three reference files, backed by numpy"""
  sys.meta_path.insert(0, _Finder)
  f = lambda fn: (lambda *a, **k: T(fn(*[np.asarray(x.a if isinstance(x, Variable) else x) for x in a], **k)))  # noqa: E731
  dtypes = module('tensorflow.python.framework.dtypes', int64=np.dtype('int64'), int32=np.dtype('int32'),
                  float32=np.dtype('float32'), bool=np.dtype('bool'), float16=np.dtype('float16'),
                  bfloat16='bfloat16', string=np.dtype('O'))
  module('tensorflow.python.framework.ops', convert_to_tensor=lambda x, name=None, dtype=None: (
      T(x.a if isinstance(x, Variable) else x, dtype)), name_scope=name_scope, device=_ctx, init_scope=_ctx,
         control_dependencies=_ctx, colocate_with=_ctx, get_default_graph=lambda: None,
         get_collection=lambda key: COLLECTIONS.setdefault(key, []),
         add_to_collection=lambda key, v: COLLECTIONS.setdefault(key, []).append(v),
         GraphKeys=types.SimpleNamespace(WEIGHTS='weights', REGULARIZATION_LOSSES='regularization_losses'))
  module('tensorflow.python.framework.sparse_tensor', SparseTensor=SparseTensor)
  module('tensorflow.python.framework.indexed_slices', IndexedSlices=IndexedSlices)
  module('tensorflow.python.framework.tensor_shape', unknown_shape=lambda n=None: TensorShape([None] * (n or 0)))
  module('tensorflow.python.framework.constant_op', constant=lambda v, **k: T(np.float32(v)))
  module('tensorflow.python.eager.context', executing_eagerly=lambda: False)

  def gather(params, indices, **k):
    p = params.a if isinstance(params, Variable) else np.asarray(params)
    return T(p[np.asarray(indices).astype(np.int64)])

  def tf_slice(x, begin, size):
    x = np.asarray(x)
    sl = tuple(slice(int(b), None if int(s) < 0 else int(b) + int(s)) for b, s in zip(begin, size))
    return T(x[sl])

  def tf_range(*a):
    return T(np.arange(*[int(x) for x in a], dtype=np.int32))

  module('tensorflow.python.ops.array_ops',
         ones_like=lambda x, dtype=None: T(np.ones_like(np.asarray(x), dtype=dtype)),
         zeros_like=f(np.zeros_like), size=lambda x: int(np.asarray(x).size), slice=tf_slice, gather=gather,
         tile=lambda x, m: T(np.tile(np.asarray(x), [int(v) for v in m])),
         reshape=lambda x, s: T(np.asarray(x).reshape([int(v) for v in np.asarray(s).reshape(-1)])),
         stack=lambda xs, axis=0: T(np.stack([np.asarray(x) for x in xs], axis=axis)),
         shape=lambda x: T(np.asarray(np.asarray(x).shape, dtype=np.int32)),
         where=lambda c, a, b, name=None: T(np.where(np.asarray(c), np.asarray(a), np.asarray(b))),
         concat=lambda xs, axis=0: T(np.concatenate([np.atleast_1d(np.asarray(x)) for x in xs], axis=axis)),
         unique=unique, expand_dims=lambda x, axis: T(np.expand_dims(np.asarray(x), axis)),
         searchsorted=lambda a, v, side='left': T(np.searchsorted(np.asarray(a), np.asarray(v), side=side)),
         split=split, squeeze=lambda x, axis=None: T(np.squeeze(np.asarray(x), axis=axis)),
         transpose=lambda x, perm=None: T(np.transpose(np.asarray(x), perm)))
  module('tensorflow.python.ops.math_ops',
         greater_equal=f(np.greater_equal), greater=f(np.greater), logical_and=f(np.logical_and),
         reduce_prod=lambda x: int(np.prod(np.asarray(x))), to_int64=lambda x: T(np.asarray(x), np.int64),
         cast=lambda x, dt: T(np.asarray(x.a if isinstance(x, Variable) else x).astype(dt)),
         segment_sum=segment_sum, div_no_nan=lambda a, b, name=None: T(np.where(np.asarray(b) == 0, 0, np.asarray(a) /
                                                                        np.where(np.asarray(b) == 0, 1, np.asarray(b)))),
         pow=f(np.power), sqrt=f(np.sqrt), sparse_segment_sum=sparse_segment('sum'),
         sparse_segment_mean=sparse_segment('mean'), sparse_segment_sqrt_n=sparse_segment('sqrtn'),
         cumsum=f(np.cumsum), range=tf_range, reduce_sum=lambda x, axis=None: T(np.sum(np.asarray(x), axis=axis)),
         add_n=lambda xs, name=None: T(sum(np.asarray(x) for x in xs)), abs=f(np.abs), multiply=lambda a, b, name=None: T(
             np.asarray(a) * np.asarray(b)))
  module('tensorflow.python.ops.sparse_ops', sparse_retain=sparse_retain, sparse_reshape=sparse_reshape,
         sparse_fill_empty_rows=sparse_fill_empty_rows)
  module('tensorflow.python.ops.embedding_ops', embedding_lookup_sparse=embedding_lookup_sparse,
         embedding_lookup=lambda params, ids, partition_strategy='mod', max_norm=None, name=None: gather(params, ids))
  module('tensorflow.python.ops.data_flow_ops', dynamic_partition=dynamic_partition,
         parallel_dynamic_stitch=parallel_dynamic_stitch)

  def scatter_update(ref, indices, updates, use_locking=False):
    ref.a[np.asarray(indices).astype(np.int64)] = np.asarray(updates, dtype=ref.a.dtype)
    return ref

  def scatter_add(ref, indices, updates, use_locking=False):
    np.add.at(ref.a, np.asarray(indices).astype(np.int64), np.asarray(updates, dtype=ref.a.dtype))
    return ref

  module('tensorflow.python.ops.state_ops', scatter_update=scatter_update, scatter_add=scatter_add)
  module('tensorflow.python.ops.control_flow_ops', group=lambda *a, **k: None)
  module('tensorflow.python.ops.resource_variable_ops')
  module('tensorflow.python.training.optimizer', Optimizer=Optimizer)
  module('tensorflow.python.training.training_ops')
  l2_loss = lambda x, name=None: T(np.sum(np.asarray(x) * np.asarray(x)) / np.asarray(x).dtype.type(2))  # noqa: E731
  module('tensorflow.python.ops.nn', l2_loss=l2_loss)
  module('tensorflow.python.ops.gen_nn_ops', l2_loss=l2_loss)
  module('tensorflow.python.ops.standard_ops', multiply=sys.modules['tensorflow.python.ops.math_ops'].multiply,
         reduce_sum=sys.modules['tensorflow.python.ops.math_ops'].reduce_sum, abs=f(np.abs))

  def clip_by_global_norm(t_list, clip_norm, use_norm=None, name=None):
    """tf.clip_by_global_norm (documented): t * clip_norm * min(1 / use_norm, 1 / clip_norm)"""
    norm = np.asarray(use_norm)
    scale = norm.dtype.type(clip_norm) * np.minimum(norm.dtype.type(1) / norm, norm.dtype.type(1) / norm.dtype.type(clip_norm))
    out = [IndexedSlices(np.asarray(t.values) * scale, t.indices) if isinstance(t, IndexedSlices) else T(np.asarray(t) * scale)
           for t in t_list]
    return out, use_norm

  module('tensorflow.python.ops.clip_ops', clip_by_global_norm=clip_by_global_norm)
  tf = module('tensorflow', constant=lambda v, **k: T(np.float32(v)))
  module('tensorflow.python.platform.tf_logging', info=lambda *a, **k: None)
  module('horovod.tensorflow', size=HVD.size, rank=HVD.rank, alltoall=HVD.alltoall, grouped_allreduce=HVD.grouped_allreduce,
         Sum=HVD.Sum, compression=HVD.compression)
  module('horovod')
  for name in ('easy_rec', 'easy_rec.python', 'easy_rec.python.compat', 'easy_rec.python.compat.feature_column',
               'easy_rec.python.compat.feature_column.utils', 'easy_rec.python.utils', 'easy_rec.python.utils.conditional',
               'easy_rec.python.utils.embedding_utils', 'easy_rec.python.compat.sok_optimizer',
               'easy_rec.python.utils.estimator_utils', 'easy_rec.python.utils.hvd_utils'):
    module(name)
  module('easy_rec.python.utils.constant', EmbeddingParallel='EmbeddingParallel')
  module('easy_rec.python.compat.dynamic_variable', DynamicVariable=type('DynamicVariable', (object,), {}))
  return tf, dtypes


COLLECTIONS = {}


def load_reference(rel_path, name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel_path))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


# ------------------------------------------------------------------------------------------------ the cases
def sparse_from_rows(rows, width=None, dtype=np.int64):
  idx = [(r, c) for r, row in enumerate(rows) for c in range(len(row))]
  vals = [v for row in rows for v in row]
  width = width or max([len(r) for r in rows] + [1])
  return SparseTensor(np.asarray(idx, dtype=np.int64).reshape(-1, 2), np.asarray(vals, dtype=dtype), [len(rows), width])


def dense_of(sp):
  """[rows, width] ids (-2 = no entry) for the fixture file"""
  out = np.full([int(d) for d in sp.dense_shape], -2, dtype=np.asarray(sp.values).dtype)
  out[tuple(np.asarray(sp.indices).T)] = np.asarray(sp.values)
  return out


def safe_lookup_cases(out, rng, emb_ops):
  table = rng.standard_normal((40, 8))
  n = 0
  for combiner in ('sum', 'mean', 'sqrtn'):
    for weighted in (False, True):
      for default_id in (None, 3):
        rows = []
        for r in range(24):
          k = int(rng.integers(0, 5)) if r % 5 else 0  # every fifth row empty
          rows.append([int(v) for v in rng.integers(-2, 40, size=k)])  # ids < 0 among them
        rows[1] = [7, 7, 7, 12]       # duplicate ids
        rows[2] = [-1, -1]            # a row that is empty AFTER pruning
        rows[3] = [5]                 # (its weight is made <= 0 below)
        ids = sparse_from_rows(rows, width=6)
        wsp, wdense = None, None
        if weighted:
          w = [[float(x) for x in rng.uniform(-0.5, 2.0, size=len(r))] for r in rows]  # some <= 0
          w[3] = [-0.25]
          w[1][0] = 0.0
          wsp = sparse_from_rows(w, width=6, dtype=np.float64)
          wdense = np.zeros([24, 6])
          wdense[tuple(np.asarray(wsp.indices).T)] = np.asarray(wsp.values)
        res = emb_ops.safe_embedding_lookup_sparse(table, ids, wsp, combiner=combiner, default_id=default_id)
        key = 'safe/%d' % n
        out[key + '/table'], out[key + '/ids'], out[key + '/out'] = table, dense_of(ids), np.asarray(res)
        out[key + '/meta'] = np.asarray([combiner, str(weighted), str(default_id)])
        if weighted:
          out[key + '/weights'] = wdense
        n += 1
  # rank-3 ids ([batch, positions, values per position]): aggregated along the last dimension
  idx = np.asarray([(0, 0, 0), (0, 0, 1), (0, 2, 0), (1, 1, 0), (1, 1, 1), (1, 1, 2), (2, 0, 0)])
  vals = np.asarray([3, 9, -1, 4, 4, 30, 11])
  sp3 = SparseTensor(idx, vals, [3, 3, 3])
  res = emb_ops.safe_embedding_lookup_sparse(table, sp3, None, combiner='mean')
  out['safe3/table'], out['safe3/indices'], out['safe3/values'], out['safe3/out'] = table, idx, vals, np.asarray(res)
  out['safe/count'] = np.asarray(n)


def ragged_cases(out, rng, fc):
  table = rng.standard_normal((30, 4))
  n = 0
  for combiner in ('sum', 'mean', 'sqrtn'):
    for weighted in (False, True):
      lens = np.asarray([3, 0, 1, 5, 2, 0, 4], dtype=np.int64)
      vals = rng.integers(0, 30, size=int(lens.sum()))
      vals[1] = vals[0]
      ids = RaggedTensor(vals, lens)
      w = RaggedTensor(rng.uniform(0.1, 2.0, size=len(vals)), lens) if weighted else None
      if weighted:
        # feature_column.py:212 expands the rank-1 flat weights at axis len(embeddings.shape) == 2: out of range for
        # tf.expand_dims as for numpy's - the weighted branch of the ragged lookup cannot run in the reference either
        try:
          fc.embedding_lookup_ragged(table, ids, w, combiner)
          raise AssertionError('the weighted ragged lookup ran')
        except np.exceptions.AxisError:
          continue
      res = np.asarray(fc.embedding_lookup_ragged(table, ids, w, combiner))
      key = 'ragged/%d' % n
      out[key + '/table'], out[key + '/values'], out[key + '/lens'], out[key + '/out'] = table, vals, lens, res
      out[key + '/meta'] = np.asarray([combiner, str(weighted)])
      if weighted:
        out[key + '/weights'] = np.asarray(w.flat_values)
      n += 1
  out['ragged/count'] = np.asarray(n)


def parallel_cases(out, rng, fc):
  """Two input forms per world size: ONE SparseTensor feature on its own table (the per-feature form; with several
  features on one table that branch concatenates row indices that restart at 0 - unsorted segment ids, which
  sparse_segment_sum rejects - so the reference's own embedding-parallel config uses the packed form), and the packed
  `sparse_fea` = (ids of N features x B rows, feature-major; lengths per (feature, row)) of
  samples/model_config/dlrm_on_criteo_parquet_ep_v2.config: N features on ONE table."""
  R, D, B = 53, 4, 6  # 53 rows: not a multiple of any world size
  table = rng.standard_normal((R, D))
  n = 0
  for world in (1, 2, 4):
    shard_rows = (R + world - 1) // world
    shards = []
    for r in range(world):  # owner r holds ids r, r + W, ...: local row = id // W   (feature_column.py:296-317, 461-463)
      sh = np.zeros((shard_rows, D))
      own = np.arange(r, R, world)
      sh[own // world] = table[own]
      shards.append(T(sh))
    for form, N in (('sparse', 1), ('packed', 3)):
      per_rank = []
      for r in range(world):  # a DIFFERENT batch per rank: multi-valued rows, empty rows, duplicate ids
        feats = []
        for _ in range(N):
          rows = [[int(v) for v in rng.integers(0, R, size=int(rng.integers(0, 4)))] for _ in range(B)]
          rows[0] = rows[0] + [rows[0][0]] if rows[0] else [1, 1]
          feats.append(rows)
        per_rank.append(feats)

      def rank_fn(rank):
        if form == 'sparse':
          lookups = [sparse_from_rows(per_rank[rank][0], width=4)]
        else:
          flat = [v for rows in per_rank[rank] for row in rows for v in row]
          lens = [len(row) for rows in per_rank[rank] for row in rows]
          lookups = {'sparse_fea': (T(flat, np.int64), T(lens, np.int64))}
        outs = [None] * N
        res = fc.embedding_parallel_lookup(shards[rank], lookups, list(range(N)), True, output_tensors=outs, batch_size=B)
        return np.asarray(res), [np.asarray(o) for o in outs]

      results = HVD.run(world, rank_fn)
      key = 'parallel/%d' % n
      out[key + '/world'], out[key + '/table'], out[key + '/features'] = np.asarray(world), table, np.asarray(N)
      for r in range(world):
        for fi in range(N):
          out['%s/rank%d/ids%d' % (key, r, fi)] = dense_of(sparse_from_rows(per_rank[r][fi], width=4))
        out['%s/rank%d/out' % (key, r)] = results[r][0]
        out['%s/rank%d/shard' % (key, r)] = np.asarray(shards[r])
        if world > 1:
          ids_sent, rows_sent = HVD.log[r][0], HVD.log[r][1]
          out['%s/rank%d/sent_ids' % (key, r)] = ids_sent['sent']
          out['%s/rank%d/sent_splits' % (key, r)] = ids_sent['splits']
          out['%s/rank%d/received_ids' % (key, r)] = ids_sent['received']
          out['%s/rank%d/sent_rows' % (key, r)] = rows_sent['sent']
      n += 1
  out['parallel/count'] = np.asarray(n)
  HVD.world = 1


def adam_cases(out, rng, adam_mod):
  """three consecutive sparse applies in float32 (de-duplicated indices, as Optimizer._apply_sparse_duplicate_indices
  hands them over), `_finish` between them"""
  f32 = np.float32
  rows, dim = 12, 4
  var = Variable((rng.standard_normal((rows, dim)) * 0.05).astype(f32), 'input_layer/C1_embedding/embedding_weights')
  opt = adam_mod.AdamOptimizerS(learning_rate=f32(0.01), beta1=f32(0.9), beta2=f32(0.999), epsilon=f32(1e-8))
  opt._create_slots([var])
  out['adam_s/var0'] = var.a.copy()
  steps = []
  for s in range(3):
    idx = np.sort(rng.choice(rows, size=5, replace=False)).astype(np.int64)
    g = (rng.standard_normal((5, dim)) * 10.0 ** rng.integers(-6, 0, size=(5, 1))).astype(f32)
    opt._lr = f32(0.01 * 0.5 ** s)
    opt._prepare()
    opt._apply_sparse_shared(T(g), var, T(idx), sys.modules['tensorflow.python.ops.state_ops'].scatter_add)
    opt._finish([], 'Adam')
    steps.append(s)
    out['adam_s/step%d/indices' % s], out['adam_s/step%d/grad' % s] = idx, g
    out['adam_s/step%d/lr' % s] = np.asarray(opt._lr, dtype=f32)
    out['adam_s/step%d/var' % s] = var.a.copy()
    out['adam_s/step%d/m' % s] = opt.get_slot(var, 'm').a.copy()
    out['adam_s/step%d/v' % s] = opt.get_slot(var, 'v').a.copy()
    out['adam_s/step%d/beta_powers' % s] = np.asarray([opt._non_slot['beta1_power'].a, opt._non_slot['beta2_power'].a], dtype=f32)
  out['adam_s/steps'] = np.asarray(len(steps))


class Weights(object):
  """a variable as a regularizer sees it: `dtype.base_dtype` and its value"""

  def __init__(self, a):
    self.a = a
    self.dtype = types.SimpleNamespace(base_dtype=a.dtype)

  def __array__(self, dtype=None, copy=None):
    return self.a if dtype is None else self.a.astype(dtype)


def regularizer_cases(out, rng, reg):
  ws = [Weights(rng.standard_normal((5, 3)).astype(np.float32)), Weights(rng.standard_normal((7,)).astype(np.float32)),
        Weights(rng.standard_normal((2, 2, 2)).astype(np.float32))]
  for i, w in enumerate(ws):
    out['reg/w%d' % i] = np.asarray(w)
  l2 = reg.l2_regularizer(1e-3)
  out['reg/l2_each'] = np.asarray([np.asarray(l2(w)) for w in ws], dtype=np.float32)
  out['reg/l2_scale'] = np.asarray(1e-3)
  assert reg.l2_regularizer(0.0)(ws[0]) is None  # "Scale of 0 disables regularizer"
  COLLECTIONS.clear()
  total = reg.apply_regularization(l2, ws)
  out['reg/apply'] = np.asarray(total, dtype=np.float32)
  assert len(COLLECTIONS['regularization_losses']) == 1  # what the estimator's add_n(REGULARIZATION_LOSSES) picks up
  both = reg.sum_regularizer([l2, reg.l2_regularizer(5e-4), reg.l2_regularizer(0.0)])
  out['reg/sum'] = np.asarray(both(ws[0]), dtype=np.float32)


def grad_norm_cases(out, rng, opt_mod):
  """`_get_grad_norm`: one dense gradient list + IndexedSlices whose values are NOT merged over duplicate indices (two
  lookups of a shared table concatenate their slices), single worker and embedding-parallel on 2 simulated ranks"""
  f32 = np.float32
  dense = [rng.standard_normal((6, 4)).astype(f32), rng.standard_normal((4,)).astype(f32)]
  for i, d in enumerate(dense):
    out['norm/dense%d' % i] = d
  # a shared table looked up twice: row 3 appears in both lookups' slices
  idx = np.asarray([3, 5, 3, 8], dtype=np.int64)
  vals = rng.standard_normal((4, 4)).astype(f32)
  out['norm/slices_indices'], out['norm/slices_values'] = idx, vals
  HVD.world = 1
  gv = [(T(d), Variable(d, 'dense%d' % i)) for i, d in enumerate(dense)] + \
       [(IndexedSlices(vals, idx), Variable(np.zeros((10, 4), f32), 'shared/embedding_weights'))]
  sparse_norm, dense_norm, grad_norm = opt_mod._get_grad_norm(gv, False)
  out['norm/single'] = np.asarray([sparse_norm, dense_norm, grad_norm], dtype=f32)
  for clip in (0.5, 100.0):
    clipped, _ = sys.modules['tensorflow.python.ops.clip_ops'].clip_by_global_norm([g for g, _ in gv], clip, use_norm=grad_norm)
    out['norm/clip%g/dense0' % clip] = np.asarray(clipped[0])
    out['norm/clip%g/slices' % clip] = np.asarray(clipped[2].values)
  # embedding-parallel, 2 ranks: each rank holds its own slices of the sharded table (named in the EmbeddingParallel
  # collection); their l2 sums are all-reduced, the dense gradients (already averaged) are identical on both ranks
  COLLECTIONS['EmbeddingParallel'] = ['sharded/embedding_weights']
  per_rank = [rng.standard_normal((3 + r, 4)).astype(f32) for r in range(2)]

  def rank_fn(rank):
    gv = [(T(d), Variable(d, 'dense%d' % i)) for i, d in enumerate(dense)] + \
         [(IndexedSlices(per_rank[rank], np.arange(len(per_rank[rank]))), Variable(np.zeros((10, 4), f32), 'sharded/embedding_weights'))]
    return [np.asarray(x) for x in opt_mod._get_grad_norm(gv, True)]

  opt_mod.hvd = sys.modules['horovod.tensorflow']
  res = HVD.run(2, rank_fn)
  for r in range(2):
    out['norm/ep/rank%d/values' % r] = per_rank[r]
    out['norm/ep/rank%d/norms' % r] = np.asarray(res[r], dtype=f32)
  HVD.world = 1


def main():
  install()
  emb_ops = load_reference('easy_rec/python/compat/embedding_ops.py', 'ref_compat_embedding_ops')
  fc = load_reference('easy_rec/python/compat/feature_column/feature_column.py', 'ref_compat_feature_column')
  fc.hvd = sys.modules['horovod.tensorflow']
  adam_mod = load_reference('easy_rec/python/compat/adam_s.py', 'ref_compat_adam_s')
  reg = load_reference('easy_rec/python/compat/regularizers.py', 'ref_compat_regularizers')
  opt_mod = load_reference('easy_rec/python/compat/optimizers.py', 'ref_compat_optimizers')
  out = {}
  rng = np.random.default_rng(20240923)
  safe_lookup_cases(out, rng, emb_ops)
  ragged_cases(out, rng, fc)
  parallel_cases(out, rng, fc)
  adam_cases(out, rng, adam_mod)
  regularizer_cases(out, rng, reg)
  grad_norm_cases(out, rng, opt_mod)
  path = os.path.join(HERE, 'embedding_stage_vectors.npz')
  np.savez_compressed(path, **out)
  print('wrote %s: %d arrays (%d safe-lookup, %d ragged, %d embedding-parallel cases)' % (
      path, len(out), int(out['safe/count']), int(out['ragged/count']), int(out['parallel/count'])))


if __name__ == '__main__':
  main()
