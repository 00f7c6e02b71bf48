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
  """farmhashna::Hash64 (every length class) in pure Python."""
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
  if n <= 64:  # farmhashna::HashLen33to64
    mul = (_K2 + n * 2) & _M
    a = (f64(0) * _K2) & _M
    b = f64(8)
    c = (f64(n - 8) * mul) & _M
    d = (f64(n - 16) * _K2) & _M
    y = (_rot((a + b) & _M, 43) + _rot(c, 30) + d) & _M
    z = _len16(y, (a + _rot((b + _K2) & _M, 18) + c) & _M, mul)
    e = (f64(16) * mul) & _M
    f = f64(24)
    g = ((y + f64(n - 32)) * mul) & _M
    h = ((z + f64(n - 24)) * mul) & _M
    return _len16((_rot((e + f) & _M, 43) + _rot(g, 30) + h) & _M, (e + _rot((f + a) & _M, 18) + g) & _M, mul)

  # farmhashna::Hash64 for more than 64 bytes: 56 bytes of state (v, w, x, y, z) over 64-byte blocks, the LAST 64 bytes of
  # the input (overlapping the final block) mixed in with a multiplier derived from the state
  def weak32(o, a, b):  # WeakHashLen32WithSeeds over the 32 bytes at offset o
    w_, x_, y_, z_ = f64(o), f64(o + 8), f64(o + 16), f64(o + 24)
    a = (a + w_) & _M
    b = _rot((b + a + z_) & _M, 21)
    c = a
    a = (a + x_ + y_) & _M
    b = (b + _rot(a, 44)) & _M
    return (a + z_) & _M, (b + c) & _M

  def shift_mix(v):
    return v ^ (v >> 47)

  seed = 81
  x = seed
  y = (seed * _K1 + 113) & _M
  z = (shift_mix((y * _K2 + 113) & _M) * _K2) & _M
  v, w = (0, 0), (0, 0)
  x = (x * _K2 + f64(0)) & _M
  end = ((n - 1) // 64) * 64
  last64 = end + ((n - 1) & 63) - 63
  assert last64 == n - 64
  o = 0
  while True:
    x = (_rot((x + y + v[0] + f64(o + 8)) & _M, 37) * _K1) & _M
    y = (_rot((y + v[1] + f64(o + 48)) & _M, 42) * _K1) & _M
    x ^= w[1]
    y = (y + v[0] + f64(o + 40)) & _M
    z = (_rot((z + w[0]) & _M, 33) * _K1) & _M
    v = weak32(o, (v[1] * _K1) & _M, (x + w[0]) & _M)
    w = weak32(o + 32, (z + w[1]) & _M, (y + f64(o + 16)) & _M)
    z, x = x, z
    o += 64
    if o == end:
      break
  mul = (_K1 + ((z & 0xff) << 1)) & _M
  o = last64
  w = ((w[0] + ((n - 1) & 63)) & _M, w[1])
  v = ((v[0] + w[0]) & _M, v[1])
  w = ((w[0] + v[0]) & _M, w[1])
  x = (_rot((x + y + v[0] + f64(o + 8)) & _M, 37) * mul) & _M
  y = (_rot((y + v[1] + f64(o + 48)) & _M, 42) * mul) & _M
  x ^= (w[1] * 9) & _M
  y = (y + v[0] * 9 + f64(o + 40)) & _M
  z = (_rot((z + w[0]) & _M, 33) * mul) & _M
  v = weak32(o, (v[1] * mul) & _M, (x + w[0]) & _M)
  w = weak32(o + 32, (z + w[1]) & _M, (y + f64(o + 16)) & _M)
  z, x = x, z
  return _len16((_len16(v[0], w[0], mul) + ((shift_mix(y) * _K0) & _M) + z) & _M, (_len16(v[1], w[1], mul) + x) & _M, mul)
