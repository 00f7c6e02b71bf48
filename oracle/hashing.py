"""ORACLE - TEST INFRASTRUCTURE ONLY.  Python access to the C restatement (farmhash_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_oracle_c.so')
_lib = None


def build(force=False):
  src = os.path.join(_HERE, 'farmhash_oracle.c')
  if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
    subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', _SO, src])
  return _SO


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = ctypes.CDLL(_SO)
    _lib.er_oracle_fingerprint64.restype = ctypes.c_uint64
    _lib.er_oracle_fingerprint64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
  return _lib


def fingerprint64(data):
  if isinstance(data, str):
    data = data.encode('utf-8')
  return int(lib().er_oracle_fingerprint64(data, len(data)))


def hash_bucket_fast(bytes_np, offsets_np, n_per_col, num_buckets, drop_empty):
  """Column-major packed strings -> int64 buckets (-1 for dropped empty strings)."""
  bytes_np = np.ascontiguousarray(bytes_np, dtype=np.uint8)
  offsets_np = np.ascontiguousarray(offsets_np, dtype=np.int64)
  nb = np.asarray(num_buckets, dtype=np.uint64).reshape(-1)
  n = len(offsets_np) - 1
  raw = bytes_np.tobytes()
  out = np.empty(n, dtype=np.int64)
  L = lib()
  for i in range(n):
    b, e = int(offsets_np[i]), int(offsets_np[i + 1])
    if e == b and drop_empty:
      out[i] = -1
    else:
      s = raw[b:e]
      out[i] = L.er_oracle_fingerprint64(s, len(s)) % int(nb[i // n_per_col])
  return out


# ---- ComboFeature through crossed_column: TF's sparse_cross_hashed (hashed_output=True) --------------------------
# reference call site: CrossedColumn._transform_feature, compat/feature_column/feature_column_v2.py:4527-4560
# (`sparse_ops.sparse_cross_hashed(inputs, num_buckets=hash_bucket_size, hash_key=self.hash_key)`), reached from
# FeatureColumnParser.parse_combo_feature (feature_column/feature_column.py:434-445, hash_key=None).  TensorFlow is
# third-party and absent from /root/reference; its published algorithm (sparse_cross_op.cc HashCrosser +
# platform/fingerprint.h FingerprintCat64): start from hash_key (default 0xDECAFCAFFE), fold every column's value -
# Fingerprint64 of a string, an int64 as it is - with FingerprintCat64, then mod num_buckets.  Pinned by the example
# in the Keras `HashedCrossing` docs (tests/test_oracle_hash.py).
DEFAULT_CROSS_HASH_KEY = 0xDECAFCAFFE


def fingerprint_cat64(fp1, fp2):
  m, k = (1 << 64) - 1, 0xc6a4a7935bd1e995

  def shift_mix(x):
    return x ^ (x >> 47)

  r = fp1 ^ k
  r ^= (shift_mix((fp2 * k) & m) * k) & m
  r = (r * k) & m
  r = (shift_mix(r) * k) & m
  return shift_mix(r)


def sparse_cross_hashed(values, num_buckets, hash_key=DEFAULT_CROSS_HASH_KEY):
  """values: one entry per crossed column (bytes / str -> Fingerprint64, int -> itself).  One combination."""
  h = hash_key
  for v in values:
    fp = (v & ((1 << 64) - 1)) if isinstance(v, (int, np.integer)) else fingerprint64(v)
    h = fingerprint_cat64(h, fp)
  return h % int(num_buckets)


def sparse_cross_hashed_columns(bytes_np, offsets_np, n_rows, n_cols, num_buckets, hash_key=DEFAULT_CROSS_HASH_KEY):
  """Column-major packed strings (string i = c * n_rows + r) -> int64 [n_rows].  '' is crossed like any value:
  CrossedColumn._transform_feature appends `inputs.get(key)` - the dense string tensor - unfiltered
  (feature_column_v2.py:4556-4558); only the hashed id columns drop '' first."""
  raw = np.ascontiguousarray(bytes_np, dtype=np.uint8).tobytes()
  off = np.ascontiguousarray(offsets_np, dtype=np.int64)
  out = np.empty(n_rows, dtype=np.int64)
  for r in range(n_rows):
    vals = [raw[int(off[c * n_rows + r]):int(off[c * n_rows + r + 1])] for c in range(n_cols)]
    out[r] = sparse_cross_hashed(vals, num_buckets, hash_key)
  return out


# ---- pure-python transcription (small cases; independent of the C file's compiler) -------------
_M = (1 << 64) - 1
_K0, _K1, _K2 = 0xc3a5c85c97cb3127, 0xb492b66fbe98f273, 0x9ae16a3b2f90404f


def _rot(v, s):
  return ((v >> s) | (v << (64 - s))) & _M if s else v


def _len16(u, v, mul):
  a = ((u ^ v) * mul) & _M
  a ^= a >> 47
  b = ((v ^ a) * mul) & _M
  b ^= b >> 47
  return (b * mul) & _M


def fingerprint64_py(s):
  """<= 32-byte branches of farmhashna::Hash64 in pure Python."""
  if isinstance(s, str):
    s = s.encode('utf-8')
  n = len(s)
  f64 = lambda o: int.from_bytes(s[o:o + 8], 'little')
  f32 = lambda o: int.from_bytes(s[o:o + 4], 'little')
  if n == 0:
    return _K2
  if n <= 3:
    y = (s[0] + (s[n >> 1] << 8)) & 0xFFFFFFFF
    z = (n + (s[n - 1] << 2)) & 0xFFFFFFFF
    v = ((y * _K2) ^ (z * _K0)) & _M
    return ((v ^ (v >> 47)) * _K2) & _M
  if n <= 7:
    mul = (_K2 + n * 2) & _M
    return _len16((n + (f32(0) << 3)) & _M, f32(n - 4), mul)
  if n <= 16:
    mul = (_K2 + n * 2) & _M
    a = (f64(0) + _K2) & _M
    b = f64(n - 8)
    c = (_rot(b, 37) * mul + a) & _M
    d = ((_rot(a, 25) + b) * mul) & _M
    return _len16(c, d, mul)
  if n <= 32:
    mul = (_K2 + n * 2) & _M
    a = (f64(0) * _K1) & _M
    b = f64(8)
    c = (f64(n - 8) * mul) & _M
    d = (f64(n - 16) * _K2) & _M
    return _len16((_rot((a + b) & _M, 43) + _rot(c, 30) + d) & _M, (a + _rot((b + _K2) & _M, 18) + c) & _M, mul)
  raise NotImplementedError('pure-python transcription covers <= 32 bytes')
