"""Host-side input parsing: raw columns -> the batch dict consumed by DeviceFeatures.load().

Mirror of the reference's `Input._preprocess` semantics (easy_rec/python/input/input.py) for the
feature types on the hot path:
  _parse_id_feature  :537-555   ids stay strings (hashed later) / string_to_number for num_buckets
  _parse_raw_feature :557-673   string_to_number, (x - min_val) / (max_val - min_val) in fp32
  _parse_tag_feature :432-505   tf.string_split(field, sep): sep is a SET of characters, empty
                                tokens skipped; kv_separator / second input column give weights
  _parse_seq_feature :677-804   tf.strings.split(field, sep): whole-string separator
  get_type_defaults  utils/input_utils.py:11-36
This is plumbing around the hot path (it runs in data-loader threads on the host); only the id
hashing is done by the library (`er_hash_bucket_fast_host` here, or on device from packed bytes).
"""
import os
import re
from collections import OrderedDict

import numpy as np

from easyrec_amd.input.features import FeatureSchema, bucketize, feature_name_of
from easyrec_amd.protos.dataset_pb2 import DatasetConfig
from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
from easyrec_amd.utils.load_class import get_register_class_meta

_INPUT_CLASS_MAP = {}
_meta_type = get_register_class_meta(_INPUT_CLASS_MAP, have_abstract_class=True)


def get_type_defaults(field_type, default_val=''):
  type_defaults = {
      DatasetConfig.INT32: 0,
      DatasetConfig.INT64: 0,
      DatasetConfig.STRING: '',
      DatasetConfig.BOOL: False,
      DatasetConfig.FLOAT: 0.0,
      DatasetConfig.DOUBLE: 0.0
  }
  assert field_type in type_defaults, 'invalid type: %s' % field_type
  if default_val == '':
    default_val = type_defaults[field_type]
  if field_type in (DatasetConfig.INT32, DatasetConfig.INT64):
    return int(default_val)
  if field_type == DatasetConfig.STRING:
    return default_val
  if field_type == DatasetConfig.BOOL:
    return str(default_val).lower() == 'true'
  return float(default_val)


class PackedCol(object):
  """One string column of a decoded text batch: cells are (begin, length) slices of the batch's text buffer
  (er_decode_csv_host) - nothing is copied until the strings are packed for hashing.  Behaves like a sequence of
  `bytes` for the generic per-row code."""

  def __init__(self, buf, begin, length):
    self.buf, self.begin, self.length = buf, begin, length

  def __len__(self):
    return len(self.begin)

  def __getitem__(self, i):
    b = int(self.begin[i])
    return self.buf[b:b + int(self.length[i])].tobytes()

  def __iter__(self):
    for i in range(len(self.begin)):
      yield self[i]


class IntCol(object):
  """An integer input column of a hashed IdFeature: hashed as its values' decimal strings (`_as_string`, input.py:356-376)."""
  __slots__ = ('values',)

  def __init__(self, values):
    self.values = np.ascontiguousarray(values, dtype=np.int64)


def as_cells(col):
  """Any string column -> (text buffer uint8, begin int64 [n], length int32 [n]): a PackedCol as it is, a list of str /
  bytes packed once (per cell, not per token)."""
  if isinstance(col, PackedCol):
    return col.buf, col.begin, col.length
  data, offsets = pack_strings(col)
  return data, offsets[:-1].copy(), np.diff(offsets).astype(np.int32)


def pack_columns(cols):
  """[PackedCol ...] over ONE text buffer -> (uint8 bytes, int64 offsets) of all their cells, column after column
  (er_pack_cells_host: one pass of memcpy instead of a Python loop over every cell)."""
  from easyrec_amd import kernels
  begin = np.concatenate([c.begin for c in cols])
  length = np.concatenate([c.length for c in cols])
  return kernels.hip().pack_cells_host(cols[0].buf, begin, length)


def pack_strings(strings):
  """list of str/bytes -> (uint8 bytes, int64 offsets[n+1])."""
  enc = [s if isinstance(s, bytes) else str(s).encode('utf-8') for s in strings]
  lens = np.fromiter((len(s) for s in enc), dtype=np.int64, count=len(enc))
  offsets = np.zeros(len(enc) + 1, dtype=np.int64)
  np.cumsum(lens, out=offsets[1:])
  data = np.frombuffer(b''.join(enc), dtype=np.uint8) if offsets[-1] > 0 else np.zeros(0, dtype=np.uint8)
  return data, offsets


def as_string(values, field_type, precision=-1):
  """`Input._as_string` (input.py:356-376): ints -> decimal, floats need an explicit precision."""
  if field_type == DatasetConfig.STRING:
    return [v if isinstance(v, str) else v.decode('utf-8') for v in values]
  if field_type in (DatasetConfig.FLOAT, DatasetConfig.DOUBLE):
    assert precision > 0, 'fc.precision not set: converting float to string is dangerous (input.py:361-367)'
    return ['%.*f' % (precision, float(v)) for v in values]
  return [str(int(v)) for v in values]


class Input(object, metaclass=_meta_type):
  """Base class: turns a dict {input_name: list/array of raw values} into a batch dict."""

  def __init__(self, data_config, feature_configs, input_path=None, batch_size=None, hash_on_host=False,
               schema_kwargs=None):
    self._data_config = data_config
    self._feature_configs = list(feature_configs)
    self._input_path = input_path
    self._batch_size = int(batch_size or data_config.batch_size)
    self._input_fields = [x.input_name for x in data_config.input_fields]
    self._input_field_types = [x.input_type for x in data_config.input_fields]
    self._input_field_defaults = [x.default_val for x in data_config.input_fields]
    self._label_fields = list(data_config.label_fields)
    self._hash_on_host = hash_on_host
    self.schema = FeatureSchema(data_config, self._feature_configs, batch_size=self._batch_size,
                                **(schema_kwargs or {}))

  def field_type(self, name):
    return self._input_field_types[self._input_fields.index(name)]

  # -- per-type parsers ---------------------------------------------------------------------
  @staticmethod
  def _to_float(col, default=0.0):
    if isinstance(col, np.ndarray) and col.dtype.kind in 'fiu':
      return col.astype(np.float32)
    out = np.empty(len(col), dtype=np.float32)
    for i, v in enumerate(col):
      if isinstance(v, bytes):
        v = v.decode('utf-8')
      out[i] = np.float32(float(v)) if v not in ('', None) else np.float32(default)
    return out

  def _parse_raw(self, fc, columns):
    x = self._to_float(columns[fc.input_names[0]])
    if fc.max_val > fc.min_val:
      # (x - min_val) / (max_val - min_val); TF casts the python floats to fp32 constants
      x = (x - np.float32(fc.min_val)) / np.float32(fc.max_val - fc.min_val)
    if fc.HasField('normalizer_fn'):
      fn = fc.normalizer_fn
      if fn in ('tf.math.log1p', 'tf.log1p'):
        x = np.log1p(x).astype(np.float32)
      else:
        raise NotImplementedError('normalizer_fn %s' % fn)
    return x.astype(np.float32)

  def _parse_raw_multi(self, fc, columns):
    """raw_input_dim > 1: 'v0<sep>v1<sep>...' -> [B, k] (input.py:569-600), then min/max normalise."""
    col = columns[fc.input_names[0]]
    k = fc.raw_input_dim
    x = np.zeros((len(col), k), dtype=np.float32)
    for i, s in enumerate(col):
      parts = self._split_charset(s, fc.separator)
      assert len(parts) <= k, 'raw feature %s: %d values > raw_input_dim %d' % (fc.input_names[0], len(parts), k)
      x[i, :len(parts)] = [np.float32(float(p)) for p in parts]
    if fc.max_val > fc.min_val:
      x = (x - np.float32(fc.min_val)) / np.float32(fc.max_val - fc.min_val)
    return x.astype(np.float32)

  @staticmethod
  def _split_charset(s, seps):
    """tf.string_split: every character of `seps` is a delimiter; empty tokens are skipped."""
    if isinstance(s, bytes):
      s = s.decode('utf-8')
    if not seps:
      return list(s)
    if len(seps) == 1:
      return [t for t in s.split(seps) if t != '']
    return [t for t in re.split('[' + re.escape(seps) + ']', s) if t != '']

  def _hash_tokens(self, tokens, buckets):
    from easyrec_amd import kernels
    data, offsets = pack_strings(tokens)
    return kernels.hip().hash_bucket_fast_host(data, offsets, max(len(tokens), 1), [buckets], False)

  # TagFeature / SequenceFeature columns split by one native pass (er_split_cells_host) instead of a Python loop over the
  # rows: hashed features with ASCII separators (the Taobao layouts); everything else keeps the per-row code below
  native_split = os.environ.get('EASYREC_AMD_NATIVE_SPLIT', '1') != '0'

  def _native_split_ok(self, fc, col):
    from easyrec_amd import kernels
    be = kernels.hip()
    sep = fc.separator
    return (self.native_split and hasattr(be, 'split_cells_host') and fc.HasField('hash_bucket_size') and
            fc.hash_bucket_size > 0 and len(sep) >= 1 and all(ord(c) < 128 for c in sep) and
            (isinstance(col, PackedCol) or all(isinstance(s, (str, bytes)) for s in col)))

  def _parse_tag(self, fc, columns, out, name):
    B = self._batch_size
    col = columns[fc.input_names[0]]
    if len(fc.input_names) == 1 and not fc.HasField('kv_separator') and self._native_split_ok(fc, col):
      from easyrec_amd import kernels
      be = kernels.hip()
      text, begin, length = as_cells(col)
      tb, tl, offs = be.split_cells_host(text, begin, length, fc.separator, keep_empty=False)
      if len(tb):
        data, offsets = be.pack_cells_host(text, tb, tl)
        ids = be.hash_bucket_fast_host(data, offsets, len(tb), [int(fc.hash_bucket_size)], False)
      else:
        ids = np.zeros(0, dtype=np.int64)
      out['tag/%s/ids' % name] = ids
      out['tag/%s/offsets' % name] = offs.astype(np.int32)
      return
    toks, offs, wts = [], np.zeros(B + 1, dtype=np.int32), []
    for i, s in enumerate(col):
      parts = self._split_charset(s, fc.separator)
      if fc.HasField('kv_separator'):
        for p in parts:
          k, v = p.split(fc.kv_separator)
          toks.append(k)
          wts.append(np.float32(float(v)))
      else:
        toks.extend(parts)
      offs[i + 1] = len(toks)
    if len(fc.input_names) > 1:
      wcol = columns[fc.input_names[1]]
      for s in wcol:
        wts.extend(np.float32(float(p)) for p in self._split_charset(s, fc.separator))
      assert len(wts) == len(toks), 'TagFeature Error: The size of %s not equal to the size of %s' % (
          fc.input_names[0], fc.input_names[1])
    if fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0:
      ids = self._hash_tokens(toks, int(fc.hash_bucket_size)) if toks else np.zeros(0, dtype=np.int64)
    elif fc.vocab_list:
      vocab = {v: i for i, v in enumerate(fc.vocab_list)}
      ids = np.array([vocab.get(t, 0) for t in toks], dtype=np.int64)
    else:
      ids = np.array([int(t) for t in toks], dtype=np.int64)
      nb = int(fc.num_buckets)
      ids = np.where((ids < 0) | (ids >= nb), 0, ids)  # IdentityCategoricalColumn default_value=0
    out['tag/%s/ids' % name] = ids
    out['tag/%s/offsets' % name] = offs
    if wts:
      out['tag/%s/weights' % name] = np.array(wts, dtype=np.float32)

  def _parse_combo_multi(self, fc, columns, out, name):
    """ComboFeature with combo_input_seps (input.py:383-407): input i is split by combo_input_seps[i] (character-set
    split, empty tokens skipped; '' = not split: the whole string, '' included, is its one value); the crossed column
    then emits one id per combination of one value from every input (no values in one input: no id for the row)."""
    from easyrec_amd import kernels
    import itertools
    B = self._batch_size
    n_in = len(fc.input_names)
    assert len(fc.combo_input_seps) == n_in, 'len(combo_separator)[%d] != len(fc.input_names)[%d]' % (
        len(fc.combo_input_seps), n_in)
    per_input = []
    for i, n in enumerate(fc.input_names):
      col, sep = columns[n], fc.combo_input_seps[i]
      if sep != '':
        per_input.append([self._split_charset(s, sep) for s in col])
      else:
        ftype = self.field_type(n)
        vals = as_string(col, ftype, fc.precision) if ftype != DatasetConfig.STRING else col
        per_input.append([[v.decode('utf-8') if isinstance(v, bytes) else v] for v in vals])
    combos, offs = [], np.zeros(B + 1, dtype=np.int32)
    for r in range(B):
      combos.extend(itertools.product(*[per_input[i][r] for i in range(n_in)]))
      offs[r + 1] = len(combos)
    n = len(combos)
    if n:
      data, offsets = pack_strings([c[i] for i in range(n_in) for c in combos])  # column-major: input i, combination k
      ids = kernels.hip().sparse_cross_hashed_host(data, offsets, n, n_in, int(fc.hash_bucket_size))
    else:
      ids = np.zeros(0, dtype=np.int64)
    out['tag/%s/ids' % name] = ids
    out['tag/%s/offsets' % name] = offs

  def _parse_lookup(self, fc, columns, out, name):
    """`Input._lookup_preprocess` (input.py:941-1000): input 0 = one key per row, input 1 = a map
    'k<kv_separator>v' joined by `separator` (character-set split, empty tokens skipped); the feature's values are the
    v of every pair whose k equals the key, in map order; they are hashed as they are ('' included: the tensor is
    already sparse when it reaches categorical_column_with_hash_bucket)."""
    B = self._batch_size
    keys, maps = columns[fc.input_names[0]], columns[fc.input_names[1]]
    max_sel = max(int(fc.lookup_max_sel_elem_num), 1)
    toks, offs = [], np.zeros(B + 1, dtype=np.int32)
    for i in range(B):
      key = keys[i].decode('utf-8') if isinstance(keys[i], bytes) else str(keys[i])
      sel = []
      for kv in self._split_charset(maps[i], fc.separator):
        parts = self._split_charset(kv, fc.kv_separator)
        assert len(parts) == 2, 'LookupFeature %s: %r is not key%svalue' % (name, kv, fc.kv_separator)
        if parts[0] == key:
          sel.append(parts[1])
      assert len(sel) <= max_sel, 'LookupFeature %s: %d values selected, lookup_max_sel_elem_num = %d' % (
          name, len(sel), max_sel)
      toks.extend(sel)
      offs[i + 1] = len(toks)
    out['tag/%s/ids' % name] = self._hash_tokens(toks, int(fc.hash_bucket_size)) if toks else np.zeros(0, dtype=np.int64)
    out['tag/%s/offsets' % name] = offs

  def _parse_seq(self, fc, columns, out, name):
    B = self._batch_size
    L = self.schema.seqs[name]['max_len']
    col = columns[fc.input_names[0]]
    ids = np.full((B, L), -1, dtype=np.int64)
    lens = np.zeros(B, dtype=np.int32)
    if 'bounds' not in self.schema.seqs[name] and len(fc.separator) == 1 and self._native_split_ok(fc, col):
      # tf.strings.split on the one-byte separator: empty tokens stay, an empty cell is one empty token, max_seq_len cut
      from easyrec_amd import kernels
      be = kernels.hip()
      text, begin, length = as_cells(col)
      tb, tl, offs = be.split_cells_host(text, begin, length, fc.separator, keep_empty=True, max_tokens=L)
      if len(tb):
        data, offsets = be.pack_cells_host(text, tb, tl)
        v = be.hash_bucket_fast_host(data, offsets, len(tb), [int(self.schema.seqs[name]['hash_buckets'])], False)
        lens[:] = np.diff(offs).astype(np.int32)
        rows = np.repeat(np.arange(B, dtype=np.int64), lens)
        ids[rows, np.arange(len(tb), dtype=np.int64) - offs[:-1][rows]] = v
      out['seq/%s/ids' % name] = ids
      out['seq/%s/len' % name] = lens
      return
    for i, s in enumerate(col):
      if isinstance(s, bytes):
        s = s.decode('utf-8')
      # tf.strings.split (input.py:685) is python's str.split with the whole separator: empty tokens stay, and an EMPTY
      # cell is one empty token.  The column takes the sparse result as it is (no ignore-value dropping for inputs
      # that are already sparse), so a hashed / vocabulary sequence embeds the empty string like any other token - an
      # empty cell is a sequence of length 1.  Number sequences cannot convert '' (TensorFlow raises): no token here.
      toks = s.split(fc.separator)
      embeds_strings = (fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0) or len(fc.vocab_list) > 0
      if toks == [''] and ('bounds' in self.schema.seqs[name] or not embeds_strings):
        toks = []
      toks = toks[:L]  # max_seq_len truncation (layers/input_layer.py:183-185)
      if not toks:
        continue
      if 'bounds' in self.schema.seqs[name]:
        # numbers (input.py:714-734: string_to_number float32, (x - min) / (max - min) with num_buckets), bucketized
        x = np.array([float(t) for t in toks], dtype=np.float32)
        if fc.num_buckets > 1 and fc.max_val > fc.min_val:
          x = (x - np.float32(fc.min_val)) / np.float32(fc.max_val - fc.min_val)
        v = bucketize(x, self.schema.seqs[name]['bounds'])
      elif fc.HasField('hash_bucket_size') and fc.hash_bucket_size > 0:
        v = self._hash_tokens(toks, int(self.schema.seqs[name]['hash_buckets']))  # (ev_params: the whole int64 range)
      elif fc.vocab_list:
        vocab = {x: j for j, x in enumerate(fc.vocab_list)}
        v = np.array([vocab.get(t, 0) for t in toks], dtype=np.int64)
      else:
        v = np.array([int(t) for t in toks], dtype=np.int64)
        nb = int(fc.num_buckets)
        v = np.where((v < 0) | (v >= nb), 0, v)
      ids[i, :len(toks)] = v
      lens[i] = len(toks)
    out['seq/%s/ids' % name] = ids
    out['seq/%s/len' % name] = lens

  # -- batch assembly -----------------------------------------------------------------------
  def preprocess(self, columns):
    """columns: {input_name: sequence of B raw values (str / bytes / numbers)} -> batch dict."""
    B = self._batch_size
    sch = self.schema
    out = OrderedDict()
    labels = np.zeros((max(len(self._label_fields), 1), B), dtype=np.float32)
    for i, name in enumerate(self._label_fields):
      labels[i] = self._to_float(columns[name])
    out['labels'] = labels
    if sch.sample_weight:
      out['sample_weight'] = self._to_float(columns[sch.sample_weight], 1.0)
    raw = np.zeros((max(sch.n_raw_rows, 1), B), dtype=np.float32)
    hash_strings = [None] * len(sch.hash_single)
    int_ids = np.zeros((max(len(sch.int_single), 1), B), dtype=np.int64)
    for fc in self._feature_configs:
      name = feature_name_of(fc)
      ft = fc.feature_type
      if ft == FeatureConfig.RawFeature:
        if name in sch.raw_multi:
          out['rawm/%s' % name] = self._parse_raw_multi(fc, columns)
        else:
          raw[sch.raw[name]['row']] = self._parse_raw(fc, columns)
          if name in sch.int_single:  # bucketized: the id is the bucket of the normalised value
            int_ids[sch.int_single[name]['col']] = bucketize(raw[sch.raw[name]['row']], sch.int_single[name]['bounds'])
      elif ft == FeatureConfig.IdFeature:
        col = columns[fc.input_names[0]]
        if name in sch.hash_single:
          ftype = self.field_type(fc.input_names[0])
          if ftype not in (DatasetConfig.FLOAT, DatasetConfig.DOUBLE) and isinstance(col, np.ndarray) and col.dtype.kind in 'iu':
            # an integer array (the Criteo binary format's categories, whatever type the data_config declares for the
            # field: a string field fed integers is hashed as str(int) too): its decimal strings are written by one
            # native pass below (IntCol), not as 4096 Python strings per feature
            hash_strings[sch.hash_single[name]['col']] = IntCol(col)
          else:
            hash_strings[sch.hash_single[name]['col']] = as_string(col, ftype, fc.precision) \
                if ftype != DatasetConfig.STRING else col
        else:
          c = sch.int_single[name]['col']
          if fc.vocab_list:
            vocab = {v: i for i, v in enumerate(fc.vocab_list)}
            int_ids[c] = [vocab.get(v if isinstance(v, str) else str(v), 0) for v in col]
          else:
            if isinstance(col, np.ndarray) and col.dtype.kind in 'iu':
              v = col.astype(np.int64)
            else:
              v = np.array([int(x) if x not in ('', b'') else 0 for x in col], dtype=np.int64)
            nb = sch.int_single[name]['num_buckets']
            int_ids[c] = np.where((v < 0) | (v >= nb), 0, v)
      elif ft == FeatureConfig.TagFeature:
        self._parse_tag(fc, columns, out, name)
      elif ft == FeatureConfig.SequenceFeature:
        self._parse_seq(fc, columns, out, name)
      elif ft == FeatureConfig.LookupFeature:
        self._parse_lookup(fc, columns, out, name)
      elif ft == FeatureConfig.ComboFeature and name in sch.tags:
        self._parse_combo_multi(fc, columns, out, name)
      elif ft == FeatureConfig.ComboFeature and name in sch.int_single:
        # crossed_column: every input as a string (input.py:407 `_as_string`), one combination per row
        strs = []
        for n in fc.input_names:
          ftype = self.field_type(n)
          col = columns[n]
          strs.extend(as_string(col, ftype, fc.precision) if ftype != DatasetConfig.STRING else col)
        data, offsets = pack_strings(strs)
        from easyrec_amd import kernels
        int_ids[sch.int_single[name]['col']] = kernels.hip().sparse_cross_hashed_host(
            data, offsets, B, len(fc.input_names), sch.int_single[name]['num_buckets'])
      elif ft == FeatureConfig.ComboFeature and name in sch.hash_single:
        # string_join of every input as a string (input.py:425-430, `_as_string`: ints in decimal, floats at fc.precision)
        cols = [as_string(columns[n], self.field_type(n), fc.precision) for n in fc.input_names]
        joined = [fc.combo_join_sep.join(c[i] for c in cols) for i in range(B)]
        hash_strings[sch.hash_single[name]['col']] = joined
    out['raw'] = raw
    out['int_ids'] = int_ids
    if hash_strings:
      if all(isinstance(s, PackedCol) for s in hash_strings) and len({id(s.buf) for s in hash_strings}) == 1:
        data, offsets = pack_columns(hash_strings)  # cells of the decoded text batch: no per-cell Python work
      elif all(isinstance(s, IntCol) for s in hash_strings):
        from easyrec_amd import kernels
        data, offsets = kernels.hip().pack_int_decimal_host(np.concatenate([s.values for s in hash_strings]))
      else:
        flat = []
        for s in hash_strings:
          if isinstance(s, IntCol):
            s = [str(int(v)) for v in s.values]
          flat.extend(s if s is not None else [''] * B)
        data, offsets = pack_strings(flat)
      if self._hash_on_host:
        from easyrec_amd import kernels
        ids = kernels.hip().hash_bucket_fast_host(data, offsets, B, sch.hash_buckets_array, True)
        out['hash_ids'] = ids.reshape(len(hash_strings), B)
      else:
        out['str_bytes'] = data
        out['str_offsets'] = offsets
    return out

  @classmethod
  def create(cls, data_config, feature_configs, input_path=None, **kwargs):
    name = DatasetConfig.InputType.Name(data_config.input_type)
    # the reader classes register themselves on import (reference: input/__init__ + utils/load_class.py)
    import importlib
    module = {'CSVInput': 'csv_input', 'ParquetInput': 'parquet_input', 'CriteoInput': 'criteo_input'}.get(name)
    if module is not None:
      importlib.import_module('easyrec_amd.input.' + module)
    return cls.create_class(name)(data_config, feature_configs, input_path, **kwargs)
