"""Schema-driven synthetic batches for any pipeline config (Taobao-shaped DIN / MMoE / DCN workloads).

Produces the packed batch dict of DeviceFeatures.load directly (no CSV round trip, ids pre-hashed):
  labels      Bernoulli per label field            raw        uniform in [min_val, max_val], normalised
  hash ids    Zipf(1.05) over the bucket range      int ids    uniform over num_buckets
  TagFeature  ragged lists of 0..max_tag ids (+weights when the feature is weighted); ~10% empty
  SequenceFeature  [B, L] ids padded with -1; lengths uniform in 1..L, the first example has length L
                   (the reference pads to the batch's longest sequence: layers/seq_input_layer.py)
"""
import numpy as np

from easyrec_amd.input.features import bucketize

from easyrec_amd.input.features import FeatureSchema, feature_name_of
from easyrec_amd.protos.feature_config_pb2 import FeatureConfig


class SyntheticBatches(object):

  def __init__(self, data_config, feature_configs, batch_size=None, seed=20240607, mode='zipf', max_tag=5,
               schema_kwargs=None):
    self.schema = FeatureSchema(data_config, feature_configs, batch_size=batch_size, **(schema_kwargs or {}))
    self.B = self.schema.batch_size
    self.rng = np.random.default_rng(seed)
    self.mode = mode
    self.max_tag = max_tag
    self.fcs = {feature_name_of(fc): fc for fc in feature_configs}
    self._cdf = {}

  def _ids(self, buckets, n):
    if self.mode == 'uniform' or buckets <= 16:
      return self.rng.integers(0, buckets, size=n, dtype=np.int64)
    V = min(buckets, 1 << 20)
    if V not in self._cdf:
      p = np.arange(1, V + 1, dtype=np.float64)**(-1.05)
      self._cdf[V] = np.cumsum(p / p.sum())
    r = np.searchsorted(self._cdf[V], self.rng.random(n)).astype(np.int64)
    return (r * 2654435761) % buckets  # spread the popular ranks over the bucket range

  def next_batch(self):
    B, sch, rng = self.B, self.schema, self.rng
    out = {}
    out['labels'] = (rng.random((max(len(sch.label_fields), 1), B)) < 0.25).astype(np.float32)
    raw = np.zeros((max(sch.n_raw_rows, 1), B), dtype=np.float32)
    for name, r in sch.raw.items():
      fc = self.fcs[name]
      lo, hi = (fc.min_val, fc.max_val) if fc.max_val > fc.min_val else (0.0, 1.0)
      x = (lo + (hi - lo) * rng.random(B)).astype(np.float32)
      if fc.max_val > fc.min_val:
        x = (x - np.float32(fc.min_val)) / np.float32(fc.max_val - fc.min_val)
      raw[r['row']] = x
    out['raw'] = raw
    for name, k in sch.raw_multi.items():
      out['rawm/%s' % name] = rng.random((B, k)).astype(np.float32)
    if sch.hash_single:
      h = np.zeros((len(sch.hash_single), B), dtype=np.int64)
      for name, info in sch.hash_single.items():
        ids = self._ids(info['buckets'], B)
        ids[rng.random(B) < 0.02] = -1  # '' -> dropped -> zero vector
        h[info['col']] = ids
      out['hash_ids'] = h
    ints = np.zeros((max(len(sch.int_single), 1), B), dtype=np.int64)
    for name, info in sch.int_single.items():
      if 'bounds' in info:  # bucketized raw feature: derived from the value drawn above
        ints[info['col']] = bucketize(raw[sch.raw[name]['row']], info['bounds'])
      else:
        ints[info['col']] = rng.integers(0, max(info['num_buckets'], 1), size=B)
    out['int_ids'] = ints
    for name, t in sch.tags.items():
      lens = rng.integers(0, self.max_tag + 1, size=B)
      lens[rng.random(B) < 0.1] = 0
      offs = np.zeros(B + 1, dtype=np.int32)
      offs[1:] = np.cumsum(lens)
      nnz = int(offs[-1])
      fc = self.fcs[name]
      buckets = t['hash_buckets'] or (len(fc.vocab_list) if fc.vocab_list else int(fc.num_buckets))
      out['tag/%s/ids' % name] = self._ids(buckets, nnz)
      out['tag/%s/offsets' % name] = offs
      if t['weighted']:
        out['tag/%s/weights' % name] = (rng.random(nnz) * 2).astype(np.float32)
    for name, s in sch.seqs.items():
      L = s['max_len']
      lens = rng.integers(1, L + 1, size=B).astype(np.int32)
      lens[0] = L
      fc = self.fcs[name]
      if 'bounds' in s:  # a sequence of numbers, bucketized (values around the boundaries' range)
        lo, hi = float(s['bounds'][0]), float(s['bounds'][-1])
        span = (hi - lo) or 1.0
        ids = bucketize((lo - 0.1 * span + 1.2 * span * rng.random(B * L)).astype(np.float32), s['bounds']).reshape(B, L)
      else:
        buckets = s['hash_buckets'] or (len(fc.vocab_list) if fc.vocab_list else int(fc.num_buckets))
        ids = self._ids(buckets, B * L).reshape(B, L)
      ids[np.arange(L)[None, :] >= lens[:, None]] = -1
      out['seq/%s/ids' % name] = ids
      out['seq/%s/len' % name] = lens
    return out
