#!/usr/bin/env python
"""Golden vectors for the packed input formats, produced by the REFERENCE's own loader code.

`easy_rec/python/input/load_parquet.py` (load_data_proc) and `easy_rec/python/input/criteo_binary_reader.py`
(BinaryDataset) import only numpy / pandas, so they run in this container: this script writes small seeded fixture
files, drives the reference functions on them (in-process, with plain queue objects in place of the multiprocessing
queues) and stores what they return in tests/golden/input_vectors.npz.  tests/test_input_formats.py re-creates the
same files from the same seed and compares easyrec_amd's readers with these vectors - /root/reference is NOT needed to
run the tests.

  python tests/golden/make_input_vectors.py           # needs /root/reference
"""
import importlib.util
import os
import queue
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/easy_rec/python/input'
BATCH = 16
PARQUET_ROWS = (37, 50, 21)        # rows per file: remainders are carried across files
CRITEO_ROWS = (40, 40, 40)         # equal parts (see the deviation note in easyrec_amd/input/criteo_input.py)
SPARSE, DENSE, LABELS = ['s1', 's2', 't1'], ['d1', 'd2'], ['label']


def write_parquet_files(dirpath, seed=20240607):
  """Three files; s1/s2 scalar int64 ids, t1 a ragged int64 list (empty rows included), d1/d2 float32, label int32."""
  import pandas as pd
  rng = np.random.default_rng(seed)
  paths = []
  for i, n in enumerate(PARQUET_ROWS):
    lens = rng.integers(0, 5, size=n)
    df = pd.DataFrame({
        'label': rng.integers(0, 2, size=n).astype(np.int32),
        's1': rng.integers(0, 10**12, size=n).astype(np.int64),
        's2': rng.integers(0, 1000, size=n).astype(np.int64),
        't1': [rng.integers(0, 10**9, size=k).astype(np.int64) for k in lens],
        'd1': rng.standard_normal(n).astype(np.float32),
        'd2': (rng.random(n) * 100).astype(np.float32),
    })
    p = os.path.join(dirpath, 'part-%d.parquet' % i)
    df.to_parquet(p)
    paths.append(p)
  return paths


def write_criteo_files(dirpath, seed=20240608):
  rng = np.random.default_rng(seed)
  out = {'label': [], 'dense': [], 'category': []}
  for i, n in enumerate(CRITEO_ROWS):
    rng.integers(0, 2, size=n).astype(np.int32).tofile(os.path.join(dirpath, 'p%d_label.bin' % i))
    rng.standard_normal((n, 13)).astype(np.float32).tofile(os.path.join(dirpath, 'p%d_dense.bin' % i))
    rng.integers(0, 2**32, size=(n, 26), dtype=np.uint64).astype(np.uint32).tofile(
        os.path.join(dirpath, 'p%d_category.bin' % i))
    for k in out:
      out[k].append(os.path.join(dirpath, 'p%d_%s.bin' % (i, k)))
  return out['label'], out['dense'], out['category']


def _load(name):
  spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, name + '.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


class _Que(object):
  """queue.Queue with the bits of multiprocessing's queue API the reference calls."""

  def __init__(self, items=()):
    self.q = queue.Queue()
    for x in items:
      self.q.put(x)

  def get(self, block=True, timeout=None):
    return self.q.get(block=block, timeout=timeout)

  def put(self, x, timeout=None):
    self.q.put(x)

  def qsize(self):
    return self.q.qsize()

  def close(self, wait_send_finish=True):
    pass


def reference_parquet_batches(paths, drop_remainder):
  lp = _load('load_parquet')
  data_que = _Que()
  cfgs = [types.SimpleNamespace(raw_input_dim=1) for _ in DENSE]
  lp.load_data_proc(0, _Que(list(paths) + [None]), data_que, _Que([True]), _Que(), BATCH, list(LABELS), list(SPARSE),
                    list(DENSE), cfgs, None, drop_remainder, 0, 1, True)
  out = []
  while data_que.qsize():
    d = data_que.get()
    if d is not None:
      out.append(d)
  return out


def reference_criteo_batches(files, rank, size, drop_last):
  br = _load('criteo_binary_reader')
  ds = br.BinaryDataset(*files, batch_size=BATCH, drop_last=drop_last, prefetch=1, global_rank=rank, global_size=size)
  return [ds[i] for i in range(len(ds))]


def main():
  store = {}
  with tempfile.TemporaryDirectory() as tmp:
    paths = write_parquet_files(tmp)
    for drop in (True, False):
      for i, d in enumerate(reference_parquet_batches(paths, drop)):
        pre = 'parquet/drop%d/%d/' % (int(drop), i)
        store[pre + 'lens'], store[pre + 'vals'] = d['sparse_fea']
        store[pre + 'dense'] = d['dense_fea']
        store[pre + 'label'] = d['label']
    files = write_criteo_files(tmp)
    for (rank, size) in ((0, 1), (0, 2), (1, 2)):
      for drop in (True, False):
        for i, (dense, cat, lbl) in enumerate(reference_criteo_batches(files, rank, size, drop)):
          pre = 'criteo/r%d_of_%d/drop%d/%d/' % (rank, size, int(drop), i)
          store[pre + 'dense'], store[pre + 'category'], store[pre + 'label'] = dense, cat, lbl
  path = os.path.join(HERE, 'input_vectors.npz')
  np.savez_compressed(path, **store)
  print('wrote %s: %d arrays, %d bytes' % (path, len(store), os.path.getsize(path)))


if __name__ == '__main__':
  sys.exit(main())
