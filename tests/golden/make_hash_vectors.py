#!/usr/bin/env python
"""Writes tests/golden/hash_vectors.json from the ORACLE (oracle/farmhash_oracle.c).

TensorFlow cannot run in this environment, so these are oracle outputs, kept so that a future
session with a TF install can diff `tf.strings.to_hash_bucket_fast` against them - in particular for
the > 16-byte branches, which have no external pin (see the oracle header).  The four 1-byte
vectors ARE externally pinned (TF's own unit test quotes their Fingerprint64 values).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import hashing  # noqa: E402


def main():
  rng = np.random.default_rng(20240607)
  items = []
  fixed = [b'', b'a', b'b', b'c', b'd', b'Hello', b'TensorFlow', b'2.x', b'68fd1e64', b'05db9164',
           b'0123456789abcdef', b'0123456789abcdefg']
  lens = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 200, 255]
  for s in fixed:
    items.append(s)
  for n in lens:
    for _ in range(2):
      items.append(bytes(rng.integers(0, 256, size=n, dtype=np.uint8)))
  vec = [{'hex': s.hex(), 'len': len(s), 'fingerprint64': str(hashing.fingerprint64(s)),
          'bucket_1e6': hashing.fingerprint64(s) % 1000000} for s in items]
  out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hash_vectors.json')
  with open(out, 'w') as f:
    json.dump({'source': 'oracle/farmhash_oracle.c (FarmHash Fingerprint64 restatement)',
               'externally_pinned': ['a', 'b', 'c', 'd', 'Hello%3', 'TensorFlow%3', '2.x%3'],
               'vectors': vec}, f, indent=1)
  print('wrote', out, len(vec))


if __name__ == '__main__':
  main()
