#!/usr/bin/env python
"""Golden vectors for the sharded embedding checkpoint files, produced by the REFERENCE's own code.

`EmbeddingParallelSaver._save_dense_embedding` / `_load_dense_embedding` (easy_rec/python/compat/
embedding_parallel_saver.py:99-168) wrap two plain-Python closures, `_save_embed` and `_load_embed` (numpy + gfile only),
into tf.py_func ops.  This script cuts those two functions (and `_get_embed_part_id`, :39-43) out of the reference's
source with `ast`, compiles them UNMODIFIED against stand-ins for what they close over (`gfile` -> os / glob / open,
`hvd.rank() / hvd.size()` -> the simulated worker, `logging`), and for a set of seeded tables
  * has W simulated workers WRITE their shards with the reference's `_save_embed` (stale parts of a larger previous world
    included: worker 0 must delete them), recording every file's bytes,
  * has the reference's `_load_embed` READ them back as worker r of W' for several (r, W'),
and stores tables, file bytes and loaded shards in tests/golden/checkpoint_vectors.npz.  tests/test_checkpoint_pins.py
holds the product's writer (er_save_dense_embed), its re-sharding loader (er_load_dense_embed = the native op
ops/src/load_dense_embed.cc) and the reader the -m gpu checkpoint test uses to these vectors; /root/reference is NOT
needed to run the tests.

  python tests/golden/make_checkpoint_vectors.py           # needs /root/reference
"""
import ast
import glob
import logging
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/easy_rec/python/compat/embedding_parallel_saver.py'

# (rows, dim, writer world, [(reader world, ...)])
CASES = [
    (103, 8, 3, (1, 2, 3, 5)),
    (64, 1, 1, (1, 4)),
    (1000, 16, 8, (1, 3, 8)),
    (17, 4, 2, (1, 2, 5)),
    (40, 16, 4, (2, 4)),
]
VAR = 'input_layer/c1_embedding/embedding_weights:0'


def _cut(tree, name):
  """the FunctionDef called `name`, wherever it is nested"""
  for node in ast.walk(tree):
    if isinstance(node, ast.FunctionDef) and node.name == name:
      return node
  raise KeyError(name)


class _GFile(object):
  """tensorflow.python.platform.gfile as the two closures use it"""

  @staticmethod
  def Exists(p):
    return os.path.exists(p)

  @staticmethod
  def MakeDirs(p):
    os.makedirs(p, exist_ok=True)

  @staticmethod
  def GFile(p, mode):
    return open(p, mode)

  @staticmethod
  def Glob(pattern):
    return glob.glob(glob.escape(os.path.dirname(pattern)) + '/' + os.path.basename(pattern).replace('[', '[[]'))

  @staticmethod
  def DeleteRecursively(p):
    os.remove(p)


def reference_functions(rank_size):
  """(_save_embed, _load_embed) compiled from the reference's source; rank_size: a mutable [rank, size] the hvd stand-in reads"""
  src = open(REF).read()
  tree = ast.parse(src)
  mod = ast.Module(body=[_cut(tree, '_get_embed_part_id'), _cut(tree, '_save_embed'), _cut(tree, '_load_embed')],
                   type_ignores=[])
  hvd = types.SimpleNamespace(rank=lambda: rank_size[0], size=lambda: rank_size[1])
  np_ns = types.SimpleNamespace(**{k: getattr(np, k) for k in ('asarray', 'zeros', 'float32', 'frombuffer', 'arange', 'where',
                                                               'logical_and', 'array', 'int64')})
  np_ns.object = object  # (`np.object` is gone from numpy 2; the closure only uses it as a dtype for the returned path)
  ns = {'np': np_ns, 'gfile': _GFile, 'hvd': hvd, 'logging': logging}
  exec(compile(mod, REF, 'exec'), ns)
  return ns['_save_embed'], ns['_load_embed']


def table_of(rows, dim):
  return np.random.default_rng(rows * 131 + dim).standard_normal((rows, dim)).astype(np.float32)


def main():
  if not os.path.exists(REF):
    sys.exit('needs /root/reference')
  rank_size = [0, 1]
  save_embed, load_embed = reference_functions(rank_size)
  out = {}
  for ci, (rows, dim, w_old, readers) in enumerate(CASES):
    table = table_of(rows, dim)
    tmp = tempfile.mkdtemp()
    try:
      ckpt = os.path.join(tmp, 'model.ckpt-7')
      n_old = (rows + w_old - 1) // w_old
      # a stale part of a larger previous world: worker 0's save must delete it
      os.makedirs(ckpt + '-embedding')
      stale = ckpt + '-embedding/embed-' + VAR.replace('/', '__') + '-part-%d.bin' % (w_old + 2)
      open(stale, 'wb').write(b'junk')
      for k in range(w_old - 1, -1, -1):   # (worker 0 last: it clears the stale parts)
        rank_size[0], rank_size[1] = k, w_old
        shard = np.zeros((n_old, dim), dtype=np.float32)   # the variable of worker k: ceil(rows / W) rows, zero padded
        mine = table[k::w_old]
        shard[:len(mine)] = mine
        save_embed(shard, ckpt.encode(), VAR.encode())
      names = sorted(os.listdir(ckpt + '-embedding'))
      assert not os.path.exists(stale), names
      out['case%d/meta' % ci] = np.array([rows, dim, w_old], dtype=np.int64)
      out['case%d/files' % ci] = np.array(names)
      for name in names:
        out['case%d/file/%s' % (ci, name)] = np.frombuffer(open(os.path.join(ckpt + '-embedding', name), 'rb').read(), np.uint8)
      for w_new in readers:
        n_new = (rows + w_new - 1) // w_new
        for r in range(w_new):
          got = load_embed(None, dim, n_new, r, w_new, ckpt.encode(), VAR.encode())
          out['case%d/load/%d_of_%d' % (ci, r, w_new)] = np.asarray(got)
    finally:
      shutil.rmtree(tmp)
  path = os.path.join(HERE, 'checkpoint_vectors.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, len(out), 'arrays', os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
