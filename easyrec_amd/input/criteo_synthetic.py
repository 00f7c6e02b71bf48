"""Synthetic Criteo-shape batches: the benchmark workload of BASELINE.json config 2.

label ~ Bernoulli(0.25); F1..F13 = min + (max-min)*u^3 inside each feature's [min_val, max_val];
C1..C26 = 8-hex-digit lowercase strings (Criteo format) drawn from per-feature vocabularies of
10^k values, k cycling 2..7, Zipf(1.05) popularity, 2% empty strings (-> zero vector); or uniform
random ids over the full vocabulary (`mode='uniform'`, worst case: nearly every lookup unique).
Batches are produced directly in the packed layout of DeviceFeatures.load (no CSV round trip).
"""
import numpy as np

from easyrec_amd.input.features import FeatureSchema, feature_name_of
from easyrec_amd.protos.feature_config_pb2 import FeatureConfig

HEX = np.frombuffer(b'0123456789abcdef', dtype=np.uint8)


def _hex8(values):
  """uint32 array -> [n, 8] ascii bytes of the zero-padded lowercase hex form."""
  v = values.astype(np.uint64)
  shifts = np.arange(28, -4, -4, dtype=np.uint64)
  nib = (v[:, None] >> shifts[None, :]) & np.uint64(0xF)
  return HEX[nib.astype(np.int64)]


_ZIPF_CDF = {}  # vocabulary size -> cumulative Zipf(1.05) distribution


class SyntheticCriteo(object):

  def __init__(self, data_config, feature_configs, batch_size=None, seed=20240607, mode='zipf',
               empty_frac=0.02):
    self.schema = FeatureSchema(data_config, feature_configs, batch_size=batch_size)
    self.B = self.schema.batch_size
    self.rng = np.random.default_rng(seed)
    self.mode = mode
    self.empty_frac = empty_frac
    self.raw_cfg = []
    for fc in feature_configs:
      if fc.feature_type == FeatureConfig.RawFeature:
        self.raw_cfg.append((feature_name_of(fc), fc))
    self.n_hash = len(self.schema.hash_single)
    self.vocab = [10**(2 + (i % 6)) for i in range(self.n_hash)]
    # a fixed random permutation-ish mixing constant per feature so that vocab ids look like hashes
    self.mix = self.rng.integers(1, 2**31 - 1, size=self.n_hash, dtype=np.int64) | 1
    self._zipf_cdf = _ZIPF_CDF  # (shared by every generator of the process: the 10^7-entry table takes seconds to build)

  def _zipf(self, V, n):
    if V not in self._zipf_cdf:
      ranks = np.arange(1, V + 1, dtype=np.float64)
      p = ranks**(-1.05)
      self._zipf_cdf[V] = np.cumsum(p / p.sum())
    u = self.rng.random(n)
    return np.searchsorted(self._zipf_cdf[V], u).astype(np.int64)

  def next_batch(self):
    B, sch = self.B, self.schema
    out = {}
    labels = (self.rng.random(B) < 0.25).astype(np.float32)
    out['labels'] = labels[None, :].copy()
    raw = np.zeros((max(sch.n_raw_rows, 1), B), dtype=np.float32)
    for name, fc in self.raw_cfg:
      u = self.rng.random(B)
      x = (fc.min_val + (fc.max_val - fc.min_val) * u**3).astype(np.float32)
      if fc.max_val > fc.min_val:
        x = (x - np.float32(fc.min_val)) / np.float32(fc.max_val - fc.min_val)
      raw[sch.raw[name]['row']] = x
    out['raw'] = raw
    out['int_ids'] = np.zeros((max(len(sch.int_single), 1), B), dtype=np.int64)
    if self.n_hash:
      n = self.n_hash * B
      vals = np.empty(n, dtype=np.int64)
      for i in range(self.n_hash):
        if self.mode == 'uniform':
          v = self.rng.integers(0, 2**32, size=B, dtype=np.int64)
        else:
          v = (self._zipf(self.vocab[i], B) * self.mix[i]) & 0xFFFFFFFF
        vals[i * B:(i + 1) * B] = v
      chars = _hex8(vals)  # [n, 8]
      empty = self.rng.random(n) < self.empty_frac
      lens = np.where(empty, 0, 8).astype(np.int64)
      offsets = np.zeros(n + 1, dtype=np.int64)
      np.cumsum(lens, out=offsets[1:])
      out['str_bytes'] = np.ascontiguousarray(chars[~empty].reshape(-1))
      out['str_offsets'] = offsets
    return out
