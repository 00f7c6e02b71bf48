"""ParquetInput: the reference's packed batch format for embedding-parallel training.

Reference: easy_rec/python/input/parquet_input.py:74-97 (which columns are sparse / dense), :201-237
(`_to_fea_dict`: ids `% num_buckets`, no raw-feature normalisation - `Input._preprocess` is bypassed, :326-329) and
easy_rec/python/input/load_parquet.py:139-317 (`load_data_proc`: the whole batches of a file are cut from its head, its last
n % B rows join the rows earlier files left over, `sparse_fea = (lens, vals)` feature-major,
`dense_fea = [B, sum raw_input_dim]`).

MI355X-first: the reference walks list columns row by row in Python (`[len(x) for x in val]`, `np.concatenate`);
here a list column is taken as Arrow's own (offsets, values) pair - already the ragged layout the lookup kernels
consume (`er_lookup_desc.offsets` / `.ids`) - so a batch is a handful of array slices, and with
`DeviceFeatures.pack()` one host-to-device copy.
"""
import glob

import numpy as np

from easyrec_amd.input.input import Input
from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
from easyrec_amd.input.features import feature_name_of


class _Column(object):
  """One input column of the rows read so far: dense values [n] / [n, k], or ragged (lens [n], vals)."""

  def __init__(self, lens, vals):
    self.lens, self.vals = lens, vals  # lens None: dense

  @staticmethod
  def from_arrow(col):
    import pyarrow as pa
    arr = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    if pa.types.is_list(arr.type) or pa.types.is_large_list(arr.type):
      offs = arr.offsets.to_numpy(zero_copy_only=False).astype(np.int64)
      vals = arr.values.to_numpy(zero_copy_only=False)
      lens = np.diff(offs).astype(np.int32)
      if arr.null_count:
        lens = np.where(arr.is_valid().to_numpy(zero_copy_only=False), lens, 0).astype(np.int32)
      return _Column(lens, vals[offs[0]:offs[-1]])
    return _Column(None, arr.to_numpy(zero_copy_only=False))

  def __len__(self):
    return len(self.vals) if self.lens is None else len(self.lens)

  def split(self, n):
    """(first n rows, the rest)"""
    if self.lens is None:
      return _Column(None, self.vals[:n]), _Column(None, self.vals[n:])
    k = int(self.lens[:n].sum())
    return _Column(self.lens[:n], self.vals[:k]), _Column(self.lens[n:], self.vals[k:])

  @staticmethod
  def concat(a, b):
    if a is None:
      return b
    if a.lens is None:
      return _Column(None, np.concatenate([a.vals, b.vals], axis=0))
    return _Column(np.concatenate([a.lens, b.lens]), np.concatenate([a.vals, b.vals]))


class ParquetInput(Input):

  def __init__(self, data_config, feature_configs, input_path=None, task_index=0, task_num=1, **kwargs):
    super(ParquetInput, self).__init__(data_config, feature_configs, input_path, **kwargs)
    self._task_index, self._task_num = task_index, task_num
    self._sparse_fcs, self._dense_fcs = [], []
    for fc in self._feature_configs:  # parquet_input.py:86-97
      if fc.feature_type in (FeatureConfig.IdFeature, FeatureConfig.TagFeature):
        self._sparse_fcs.append(fc)
      elif fc.feature_type == FeatureConfig.RawFeature:
        self._dense_fcs.append(fc)
      else:
        raise AssertionError('feature_type[%s] not supported' % str(fc.feature_type))
    # every id column shares one modulus (parquet_input.py:211-221)
    nb = {int(fc.num_buckets) for fc in self._feature_configs if fc.num_buckets > 0}
    assert len(nb) <= 1, 'all features must share the same buckets, but are %s' % sorted(nb)
    self._num_buckets = nb.pop() if nb else -1

  # -- files -----------------------------------------------------------------------------------
  def files(self):
    paths = self._input_path if isinstance(self._input_path, (list, tuple)) else str(self._input_path).split(',')
    out = []
    for p in paths:
      out.extend(sorted(glob.glob(p)) or [p])
    # worker w takes files w, w + task_num, ... (parquet_input.py:52-57)
    return [f for i, f in enumerate(out) if i % self._task_num == self._task_index]

  def _fields(self):
    names = [fc.input_names[0] for fc in self._sparse_fcs] + [fc.input_names[0] for fc in self._dense_fcs]
    return names + [x for x in self._label_fields if x not in names]

  def _read(self, path):
    import pyarrow.parquet as pq
    table = pq.read_table(path, columns=self._fields())
    return {name: _Column.from_arrow(table.column(name)) for name in table.column_names}

  # -- the reference's packed batches (load_parquet.py) -----------------------------------------
  def reference_batches(self, drop_remainder=True, num_epochs=1):
    """Yields what `load_data_proc(..., need_pack=True)` puts on its queue for ONE data process: dicts with
    'sparse_fea' = (lens int32 [F * B], vals [sum lens]) feature-major, 'dense_fea' [B, sum raw_input_dim],
    and one array per label field."""
    B = self._batch_size
    sparse = [fc.input_names[0] for fc in self._sparse_fcs]
    dense = [fc.input_names[0] for fc in self._dense_fcs]

    def emit(cols):
      d = {}
      if sparse:
        parts = [cols[k] if cols[k].lens is not None else
                 _Column(np.ones(len(cols[k]), dtype=np.int32), cols[k].vals) for k in sparse]
        d['sparse_fea'] = (np.concatenate([p.lens for p in parts]), np.concatenate([p.vals for p in parts]))
      if dense:
        d['dense_fea'] = np.concatenate(
            [np.asarray(cols[k].vals).reshape(-1, max(int(fc.raw_input_dim), 1)) for k, fc in zip(dense, self._dense_fcs)],
            axis=1)
      for k in self._label_fields:
        d[k] = np.asarray(cols[k].vals)
      return d

    for _ in range(num_epochs):
      carry = None
      for path in self.files():
        cols = self._read(path)
        n = len(next(iter(cols.values())))
        # load_parquet.py:170-221: the whole batches of a file come first, from its own head ...
        for _b in range(n // B):
          head = {}
          for k in list(cols):
            head[k], cols[k] = cols[k].split(B)
          yield emit(head)
        # ... :223-287: then its last n % B rows are appended to what earlier files left over
        if n % B > 0:
          if carry is not None:
            cols = {k: _Column.concat(carry[k], v) for k, v in cols.items()}
          if len(next(iter(cols.values()))) >= B:
            head = {}
            for k in list(cols):
              head[k], cols[k] = cols[k].split(B)
            yield emit(head)
          carry = cols if len(next(iter(cols.values()))) > 0 else None
      if carry is not None and not drop_remainder:
        yield emit(carry)

  # -- batches of this framework ----------------------------------------------------------------
  def from_reference_batch(self, d):
    """Packed reference batch -> batch dict of DeviceFeatures.load (full batches only: the device buffers are
    sized for data_config.batch_size)."""
    B, sch = self._batch_size, self.schema
    out = {}
    labels = np.zeros((max(len(self._label_fields), 1), B), dtype=np.float32)
    for i, k in enumerate(self._label_fields):
      assert len(d[k]) == B, 'ParquetInput: %d rows in a batch of %d (use drop_remainder)' % (len(d[k]), B)
      labels[i] = np.asarray(d[k], dtype=np.float32).reshape(-1)
    out['labels'] = labels
    raw = np.zeros((max(sch.n_raw_rows, 1), B), dtype=np.float32)
    if self._dense_fcs:
      col = 0
      for fc in self._dense_fcs:
        k, name = max(int(fc.raw_input_dim), 1), feature_name_of(fc)
        block = d['dense_fea'][:, col:col + k].astype(np.float32)
        col += k
        if name in sch.raw_multi:
          out['rawm/%s' % name] = block
        else:
          raw[sch.raw[name]['row']] = block[:, 0]
    out['raw'] = raw
    int_ids = np.zeros((max(len(sch.int_single), 1), B), dtype=np.int64)
    if self._sparse_fcs:
      lens, vals = d['sparse_fea']
      vals = np.asarray(vals).astype(np.int64)
      if self._num_buckets > 0:
        vals = vals % self._num_buckets  # parquet_input.py:222 (numpy's % is floored like tf's)
      pos = 0
      for f, fc in enumerate(self._sparse_fcs):
        name = feature_name_of(fc)
        fl = lens[f * B:(f + 1) * B]
        n = int(fl.sum())
        fv = vals[pos:pos + n]
        pos += n
        if name in sch.int_single:
          assert n == B and (fl == 1).all(), 'IdFeature %s: exactly one id per row expected' % name
          int_ids[sch.int_single[name]['col']] = fv
        elif name in sch.tags:
          cap = sch.tags[name]['cap']
          assert n <= cap, 'tag feature %s: %d ids exceed capacity %d' % (name, n, cap)
          offsets = np.zeros(B + 1, dtype=np.int32)
          np.cumsum(fl, out=offsets[1:])
          out['tag/%s/ids' % name] = fv
          out['tag/%s/offsets' % name] = offsets
          if sch.tags[name]['weighted']:
            out['tag/%s/weights' % name] = np.ones(n, dtype=np.float32)
        else:
          raise AssertionError('ParquetInput: feature %s needs num_buckets (ids are not hashed)' % name)
    out['int_ids'] = int_ids
    return out

  def batches(self, num_epochs=None, drop_remainder=True):
    assert drop_remainder, 'ParquetInput: the device buffers hold full batches only'
    for d in self.reference_batches(drop_remainder=True, num_epochs=num_epochs or self._data_config.num_epochs or 1):
      yield self.from_reference_batch(d)
