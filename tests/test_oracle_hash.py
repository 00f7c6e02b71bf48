"""Pins the hashing oracle to TensorFlow's published vectors, then the library's host entry point
(er_hash_bucket_fast_host) to the oracle."""
import ctypes
import json
import os

import numpy as np

from oracle import hashing

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'hash_vectors.json')


def test_tf_fingerprint64_vectors():
  # quoted in tensorflow/python/kernel_tests/string_to_hash_bucket_op_test.py (testStringToHashBucketsFast)
  known = {'a': 12917804110809363939, 'b': 11795596070477164822, 'c': 11430444447143000872,
           'd': 4470636696479570465}
  for s, h in known.items():
    assert hashing.fingerprint64(s) == h
    assert hashing.fingerprint64_py(s) == h
  assert [hashing.fingerprint64(s) % 10 for s in 'abcd'] == [9, 2, 2, 5]


def test_tf_docs_example():
  # tf.strings.to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) -> [0, 2, 2]
  assert [hashing.fingerprint64(s) % 3 for s in ['Hello', 'TensorFlow', '2.x']] == [0, 2, 2]


def test_c_matches_python_transcription():
  """Every length class of Fingerprint64 - 0-16, 17-32, 33-64 bytes and the 64-byte-block loop with each tail length
  (65 ... 300: one to four blocks + every remainder) - through BOTH restatements (the C oracle and the Python
  transcription, typed separately from the published algorithm); the inputs above 16 bytes have no external vector."""
  rng = np.random.default_rng(0)
  for n in list(range(0, 301)):
    for _ in range(20 if n <= 32 else 6):
      s = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
      assert hashing.fingerprint64(s) == hashing.fingerprint64_py(s), n
  for n in (511, 512, 513, 1000, 4096):
    s = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    assert hashing.fingerprint64(s) == hashing.fingerprint64_py(s), n


def test_golden_file_matches():
  """tests/golden/hash_vectors.json (written by tests/golden/make_hash_vectors.py) is reproduced."""
  with open(GOLDEN) as f:
    vec = json.load(f)
  for item in vec['vectors']:
    assert hashing.fingerprint64(bytes.fromhex(item['hex'])) == int(item['fingerprint64'])


def _random_strings(rng, n, maxlen):
  out = []
  for _ in range(n):
    ln = int(rng.integers(0, maxlen + 1))
    out.append(bytes(rng.integers(0, 256, size=ln, dtype=np.uint8)))
  return out


def test_library_host_hash_matches_oracle(built_lib):
  from easyrec_amd import kernels
  from easyrec_amd.input.input import pack_strings
  be = kernels.HipBackend()
  rng = np.random.default_rng(1)
  strs = _random_strings(rng, 3000, 200) + [b'', b'a', b'x' * 64, b'y' * 65, b'z' * 128, b'w' * 129]
  n = len(strs) // 2 * 2
  strs = strs[:n]
  data, offs = pack_strings(strs)
  nb = np.array([1000000, 7], dtype=np.uint64)
  for drop in (0, 1):
    got = be.hash_bucket_fast_host(data, offs, n // 2, nb, drop)
    exp = hashing.hash_bucket_fast(data, offs, n // 2, nb, drop)
    assert np.array_equal(got, exp)


def test_sparse_cross_hashed_known_answer_from_the_keras_docs():
  """tf.keras.layers.HashedCrossing(num_bins=5) on (['A','B','A','B','A'], [101,101,101,102,102]) -> [1, 4, 1, 1, 3]
  (TF API docs; the layer calls tf.sparse.cross_hashed(inputs, num_bins): strings are fingerprinted, int64 values go
  in as they are, folded with FingerprintCat64 from the default hash_key 0xDECAFCAFFE).  This pins the restatement
  behind ComboFeature / crossed_column (reference feature_column/feature_column.py:434-445)."""
  from oracle import hashing
  f1, f2 = ['A', 'B', 'A', 'B', 'A'], [101, 101, 101, 102, 102]
  assert [hashing.sparse_cross_hashed([a, b], 5) for a, b in zip(f1, f2)] == [1, 4, 1, 1, 3]
  # order matters (the fold is not symmetric), and the key is part of the hash
  assert hashing.sparse_cross_hashed(['A', 'B'], 1 << 40) != hashing.sparse_cross_hashed(['B', 'A'], 1 << 40)
  assert hashing.sparse_cross_hashed(['A', 'B'], 1 << 40) != hashing.sparse_cross_hashed(['A', 'B'], 1 << 40, hash_key=1)


def test_sparse_cross_hashed_host_entry_equals_the_oracle(built_lib):
  """er_sparse_cross_hashed_host (the product's input-stage entry point) against the restatement: random strings of
  0..40 bytes in 2 and 3 columns ('' included: the crossed column does not filter its inputs)."""
  from easyrec_amd.input.input import pack_strings
  rng = np.random.default_rng(5)
  for n_cols, n_rows, buckets in ((2, 300, 1000), (3, 257, 1 << 33)):
    strs = []
    for _ in range(n_cols * n_rows):
      n = int(rng.integers(0, 41)) if rng.random() > 0.1 else 0
      strs.append(bytes(rng.integers(1, 256, size=n, dtype=np.uint8)))
    data, offsets = pack_strings(strs)
    exp = hashing.sparse_cross_hashed_columns(data, offsets, n_rows, n_cols, buckets)
    lib = ctypes.CDLL(built_lib)
    out = np.empty(n_rows, dtype=np.int64)
    d = np.ascontiguousarray(data, dtype=np.uint8)
    o = np.ascontiguousarray(offsets, dtype=np.int64)
    rc = lib.er_sparse_cross_hashed_host(d.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.c_int64(n_rows), ctypes.c_int32(n_cols), ctypes.c_uint64(buckets),
                                         ctypes.c_uint64(hashing.DEFAULT_CROSS_HASH_KEY), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.array_equal(out, exp)
    assert (out >= 0).all()  # '' is crossed like any other value


def test_fingerprint64_bigquery_documentation_vectors():
  """FARM_FINGERPRINT is farmhash Fingerprint64 (as a signed int64); the example in BigQuery's function reference
  fingerprints CONCAT(x, y, z) of the rows (1, "foo", true), (2, "apple", false), (3, "", true): full 64-bit known
  answers for inputs of 8, 11 and 5 bytes."""
  from oracle import hashing
  for s, want in ((b'1footrue', -1541654101129638711), (b'2applefalse', 2794438866806483259), (b'3true', -4880158226897771312)):
    for fn in (hashing.fingerprint64, hashing.fingerprint64_py):
      v = int(fn(s))
      assert (v - (1 << 64) if v >= (1 << 63) else v) == want, (s, fn.__name__)
