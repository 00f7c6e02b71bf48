#!/usr/bin/env python
"""What the REFERENCE'S OWN input preprocessing (easy_rec/python/input/input.py: Input._parse_id_feature :537-555,
_parse_raw_feature :557-673, _parse_tag_feature :432-505, _parse_seq_feature :677-760, _parse_combo_feature :378-430,
_lookup_preprocess :941-1003, _as_string :356-376) makes of raw
columns - run in the build container where /root/reference exists.

Those methods are Python over a dozen TensorFlow string / sparse ops.  This script executes them, unmodified, on a numpy
stand-in that implements the documented semantics of those ops:
  tf.string_split(source, delimiter, skip_empty=True)  every CHARACTER of `delimiter` separates; empty tokens dropped
  tf.strings.split(source, sep)                        python's str.split(sep): the whole `sep`, empty tokens kept,
                                                       '' -> ['']
  tf.string_to_number, tf.as_string (integers; floats with `precision` digits), tf.sparse_to_dense, SparseTensor
and stores, per feature, a canonical form of the result (tests/golden/preprocess_vectors.json): strings per row for the
hashed id columns, integers for the num_buckets ones, float32 values for raw features, ragged token / weight lists per row
for tags and sequences.  tests/test_preprocess_pins.py feeds the same raw columns to easyrec_amd/input/input.py and
compares (hashed columns through the oracle's pinned Fingerprint64).

usage: python tests/golden/make_preprocess_vectors.py [/root/reference]
"""
import contextlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


class Shape(tuple):

  def as_list(self):
    return list(self)


class T(np.ndarray):
  """numeric tensor: a numpy array that answers get_shape()"""

  def get_shape(self):
    return Shape(np.ndarray.shape.__get__(self))


def num(x, dtype=None):
  return np.asarray(x, dtype=dtype).view(T)


class Str(object):
  """string tensor (object array of python str)"""
  dtype = 'string'

  def __init__(self, values):
    self.a = np.asarray(values, dtype=object)

  def get_shape(self):
    return Shape(self.a.shape)

  @property
  def shape(self):
    return Shape(self.a.shape)

  def __getitem__(self, i):
    r = self.a[i]
    return Str(r) if isinstance(r, np.ndarray) else Scalar(r)


class Scalar(str):
  """one string of a string tensor, as tf.map_fn hands it to its function"""
  dtype = 'string'

  def get_shape(self):
    return Shape(())

  @property
  def a(self):
    return np.asarray(str(self), dtype=object)


class Sparse(object):

  def __init__(self, indices, values, dense_shape):
    self.indices = num(np.asarray(indices, dtype=np.int64).reshape(-1, len(dense_shape) if len(np.shape(indices)) < 2
                                                                     else np.shape(indices)[1]))
    self.values = values
    self.dense_shape = num(dense_shape, np.int64)

  def rows(self):
    """ragged python lists per first-dimension row, in index order"""
    vals = self.values.a if isinstance(self.values, Str) else np.asarray(self.values)
    out = [[] for _ in range(int(self.dense_shape[0]))]
    for idx, v in zip(np.asarray(self.indices), vals):
      out[int(idx[0])].append(v if isinstance(v, str) else v.item())
    return out


def _tokens_charset(s, delims, skip_empty):
  if delims == '':
    toks = list(s)
  else:
    toks, cur = [], ''
    for ch in s:
      if ch in delims:
        toks.append(cur)
        cur = ''
      else:
        cur += ch
    toks.append(cur)
  return [t for t in toks if t != ''] if skip_empty else toks


def _split(source, tokens_of):
  src = source.a if isinstance(source, Str) else np.asarray(source, dtype=object)
  idx, vals, width = [], [], 0
  for r, s in enumerate(src.reshape(-1)):
    toks = tokens_of(s)
    width = max(width, len(toks))
    for c, tok in enumerate(toks):
      idx.append((r, c))
      vals.append(tok)
  return Sparse(np.asarray(idx, dtype=np.int64).reshape(-1, 2), Str(vals), [src.size, width])


def make_tf():
  tf = types.ModuleType('tensorflow')
  tf.__version__ = '1.15.0'
  tf.string, tf.float32, tf.double, tf.float64, tf.int32, tf.int64, tf.bool = 'string', np.float32, np.float64, np.float64, \
      np.int32, np.int64, np.bool_
  tf.string_split = lambda source, delimiter=' ', skip_empty=True: _split(source, lambda s: _tokens_charset(s, delimiter, skip_empty))
  tf.strings = types.SimpleNamespace(
      split=lambda source, sep=None: _split(source, lambda s: s.split(sep)),
      as_string=lambda x, precision=None: as_string(x, precision))
  tf.as_string = tf.strings.as_string

  def as_string(x, precision=None):
    a = np.asarray(x)
    if a.dtype.kind in 'iu':
      return Str([str(int(v)) for v in a.reshape(-1)])
    assert precision is not None
    return Str([('%.' + str(precision) + 'f') % float(v) for v in a.reshape(-1)])

  def string_to_number(x, out_type=np.float32, name=None):
    src = x.a if isinstance(x, Str) else np.asarray(x, dtype=object)
    if out_type in (np.int32, np.int64):
      return num([int(s) for s in src.reshape(-1)], out_type).reshape(src.shape)
    return num([np.float32(s) if out_type == np.float32 else float(s) for s in src.reshape(-1)], out_type).reshape(src.shape)

  tf.string_to_number = string_to_number
  tf.sparse = types.SimpleNamespace(SparseTensor=Sparse)
  tf.SparseTensor = lambda indices, values, dense_shape: Sparse(indices, values, [int(d) for d in np.asarray(dense_shape).reshape(-1)])
  tf.expand_dims = lambda x, axis=0: Str(np.expand_dims(x.a, axis)) if isinstance(x, Str) else num(np.expand_dims(x, axis))
  tf.squeeze = lambda x, axis=None: Str(np.squeeze(x.a, axis)) if isinstance(x, Str) else num(np.squeeze(x, axis))

  def reshape(x, shape, name=None):
    shape = [int(s) for s in shape]
    return Str(x.a.reshape(shape)) if isinstance(x, Str) else num(np.reshape(x, shape))

  tf.reshape = reshape
  tf.shape = lambda x: num((x.a if isinstance(x, Str) else np.asarray(x)).shape, np.int64)
  tf.range = lambda *a, **k: num(np.arange(*[int(v) for v in a], dtype=k.get('dtype', np.int64)))
  tf.tile = lambda x, m: num(np.tile(np.asarray(x), [int(v) for v in m]))
  tf.concat = lambda xs, axis=0: num(np.concatenate([np.asarray(x) for x in xs], axis=axis))
  tf.to_int64 = lambda x: num(x, np.int64)
  tf.to_float = lambda x: num(x, np.float32)
  tf.cast = lambda x, d: num(x, d)
  tf.identity = lambda x: x
  tf.gather = lambda x, i: num(np.asarray(x)[np.asarray(i)])
  tf.assert_equal = lambda a, b, message=None: None
  tf.control_dependencies = lambda ops: contextlib.nullcontext()
  tf.py_func = lambda *a, **k: None

  def sparse_to_dense(indices, shape, values, default_value=0):
    out = np.full([int(s) for s in shape], default_value, dtype=np.asarray(values).dtype)
    for idx, v in zip(np.asarray(indices), np.asarray(values)):
      out[tuple(int(i) for i in idx)] = v
    return num(out)

  tf.sparse_to_dense = sparse_to_dense
  # the ops of Input._lookup_preprocess (input.py:941-1003) and _parse_combo_feature (:378-430)
  tf.equal = lambda a, b: (a.a if isinstance(a, Str) else np.asarray(a)) == (b.a if isinstance(b, Str) else b)
  tf.where = lambda cond: num(np.argwhere(np.asarray(cond)), np.int64)
  tf.gather = lambda x, i: Str(x.a[np.asarray(i, dtype=np.int64)]) if isinstance(x, Str) else num(np.asarray(x)[np.asarray(i)])
  tf.pad = lambda x, paddings: Str(np.concatenate([x.a, np.array([''] * int(paddings[0][1]), dtype=object)])) \
      if isinstance(x, Str) else num(np.pad(np.asarray(x), [tuple(int(v) for v in p) for p in paddings]))
  tf.sequence_mask = lambda n, maxlen: np.arange(int(maxlen)) < int(n)
  tf.zeros = lambda shape, dtype=np.float32: num(np.zeros([int(v) for v in shape], dtype=dtype))
  tf.stack = lambda xs: num([int(v) for v in xs], np.int64)
  tf.reduce_max = lambda x: np.max(np.asarray(x))

  def map_fn(fn, elems, dtype=None):
    outs = [fn([e[i] if isinstance(e, Str) else np.asarray(e)[i] for e in elems]) for i in range(len(elems[0].a))]
    cols = list(zip(*outs))
    return tuple(Str(np.stack([c.a for c in col])) if isinstance(col[0], Str) else num(np.stack([np.asarray(c) for c in col]))
                 for col in cols)

  tf.map_fn = map_fn
  tf.boolean_mask = lambda x, m: Str(x.a[np.asarray(m)]) if isinstance(x, Str) else num(np.asarray(x)[np.asarray(m)])
  tf.compat = types.SimpleNamespace(v1=tf)
  return tf


def canonical(v):
  if isinstance(v, Sparse):
    return {'sparse_rows': v.rows(), 'dense_shape': [int(d) for d in v.dense_shape]}
  if isinstance(v, Str):
    return {'strings': [str(s) for s in v.a.reshape(-1)]}
  a = np.asarray(v)
  return {'dtype': str(a.dtype), 'shape': list(a.shape), 'values': [x.item() for x in a.reshape(-1)]}


def main():
  from google.protobuf import text_format

  from easyrec_amd import protos
  import preprocess_cases as pc
  sys.modules['tensorflow'] = make_tf()
  stubs = ('tensorflow.python', 'tensorflow.python.framework', 'tensorflow.python.framework.ops', 'tensorflow.python.ops',
           'tensorflow.python.ops.array_ops', 'tensorflow.python.ops.sparse_ops', 'tensorflow.python.ops.string_ops',
           'tensorflow.python.platform', 'tensorflow.python.platform.gfile', 'easy_rec', 'easy_rec.python', 'easy_rec.python.core',
           'easy_rec.python.core.sampler', 'easy_rec.python.protos', 'easy_rec.python.utils', 'easy_rec.python.utils.conditional',
           'easy_rec.python.utils.config_util', 'easy_rec.python.utils.constant', 'easy_rec.python.utils.check_utils',
           'easy_rec.python.utils.expr_util', 'easy_rec.python.utils.input_utils', 'easy_rec.python.utils.load_class',
           'easy_rec.python.utils.tf_utils')
  for name in stubs:
    sys.modules[name] = types.ModuleType(name)
  for parent, child in (('tensorflow.python.framework', 'ops'), ('tensorflow.python.ops', 'array_ops'),
                        ('tensorflow.python.ops', 'sparse_ops'), ('tensorflow.python.ops', 'string_ops'),
                        ('tensorflow.python.platform', 'gfile'), ('easy_rec.python.core', 'sampler'),
                        ('easy_rec.python.utils', 'conditional'), ('easy_rec.python.utils', 'config_util'),
                        ('easy_rec.python.utils', 'constant')):
    setattr(sys.modules[parent], child, sys.modules[parent + '.' + child])
  sys.modules['easy_rec.python.protos.dataset_pb2'] = protos.dataset_pb2
  sys.modules['tensorflow.python.ops.string_ops'].string_join = lambda inputs, separator='': Str(
      [separator.join(str(i.a.reshape(-1)[r]) for i in inputs) for r in range(inputs[0].a.size)])
  sys.modules['easy_rec.python.utils.check_utils'].check_split = None
  sys.modules['easy_rec.python.utils.check_utils'].check_string_to_number = None
  sys.modules['easy_rec.python.utils.expr_util'].get_expression = None
  sys.modules['easy_rec.python.utils.input_utils'].get_type_defaults = None
  sys.modules['easy_rec.python.utils.load_class'].get_register_class_meta = lambda *a, **k: type
  sys.modules['easy_rec.python.utils.load_class'].load_by_path = None
  sys.modules['easy_rec.python.utils.tf_utils'].get_tf_type = None
  spec = importlib.util.spec_from_file_location('ref_input', os.path.join(REF, 'easy_rec/python/input/input.py'))
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)

  data_config = protos.dataset_pb2.DatasetConfig()
  text_format.Merge(pc.DATA_CONFIG, data_config)
  features = protos.feature_config_pb2.FeatureConfigV2()
  text_format.Merge(pc.FEATURES, features)
  inp = object.__new__(ref.Input)
  inp._check_mode, inp._data_config, inp._normalizer_fn, inp._appended_fields = False, data_config, {}, []
  kinds = {f.input_name: f.input_type for f in data_config.input_fields}
  field_dict = {}
  for name, col in pc.COLUMNS.items():
    t = kinds[name]
    if t == protos.dataset_pb2.DatasetConfig.STRING:
      field_dict[name] = Str(col)
    elif t in (protos.dataset_pb2.DatasetConfig.INT32, protos.dataset_pb2.DatasetConfig.INT64):
      field_dict[name] = num(col, np.int64 if t == protos.dataset_pb2.DatasetConfig.INT64 else np.int32)
    else:
      field_dict[name] = num(col, np.float32 if t == protos.dataset_pb2.DatasetConfig.FLOAT else np.float64)
  FC = protos.feature_config_pb2.FeatureConfig
  parsed = {}
  for fc in features.features:
    if fc.feature_type == FC.LookupFeature:  # (as Input._preprocess dispatches it: input.py:851-855)
      parsed[fc.feature_name] = inp._lookup_preprocess(fc, field_dict)
      continue
    {FC.IdFeature: inp._parse_id_feature, FC.RawFeature: inp._parse_raw_feature, FC.TagFeature: inp._parse_tag_feature,
     FC.SequenceFeature: inp._parse_seq_feature, FC.ComboFeature: inp._parse_combo_feature}[fc.feature_type](fc, parsed, field_dict)
  out = {'generator': 'tests/golden/make_preprocess_vectors.py', 'parsed': {k: canonical(v) for k, v in parsed.items()}}
  path = os.path.join(HERE, 'preprocess_vectors.json')
  with open(path, 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print('wrote %s: %d parsed entries: %s' % (path, len(parsed), ', '.join(sorted(parsed))))


if __name__ == '__main__':
  main()
