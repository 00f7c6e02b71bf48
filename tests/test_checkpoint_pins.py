"""The sharded embedding checkpoint files against the REFERENCE's own writer and reader (tests/golden/
checkpoint_vectors.npz: `_save_embed` / `_load_embed` of compat/embedding_parallel_saver.py:99-168 executed by
tests/golden/make_checkpoint_vectors.py), and the tensor-bundle files against an independent reader."""
import os

import numpy as np
import pytest

from _ckpt_readers import crc32c, read_bundle, ref_load_embed

HERE = os.path.dirname(os.path.abspath(__file__))
VAR = 'input_layer/c1_embedding/embedding_weights:0'
FILE_VAR = 'embed-' + VAR.replace('/', '__')


def _cases():
  z = np.load(os.path.join(HERE, 'golden', 'checkpoint_vectors.npz'))
  n = len([k for k in z.files if k.endswith('/meta')])
  return z, n


def _table_of(rows, dim):  # (make_checkpoint_vectors.table_of)
  return np.random.default_rng(rows * 131 + dim).standard_normal((rows, dim)).astype(np.float32)


@pytest.mark.parametrize('ci', range(5))
def test_writer_and_loaders_against_the_reference_functions(tmp_path, built_lib, ci):
  from easyrec_amd import kernels
  be = kernels.HipBackend()
  z, n = _cases()
  assert n == 5
  rows, dim, w_old = [int(x) for x in z['case%d/meta' % ci]]
  table = _table_of(rows, dim)
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-7')
  n_old = (rows + w_old - 1) // w_old
  os.makedirs(ckpt + '-embedding')
  open(os.path.join(ckpt + '-embedding', FILE_VAR + '-part-%d.bin' % (w_old + 2)), 'wb').write(b'junk')
  for k in range(w_old - 1, -1, -1):
    shard = np.zeros((n_old, dim), dtype=np.float32)
    mine = table[k::w_old]
    shard[:len(mine)] = mine
    be.save_dense_embed(ckpt, FILE_VAR, k, w_old, shard)
  # 1. the files er_save_dense_embed writes are, byte for byte, the files the reference's _save_embed wrote
  names = sorted(os.listdir(ckpt + '-embedding'))
  assert names == [str(x) for x in z['case%d/files' % ci]]
  for name in names:
    got = np.frombuffer(open(os.path.join(ckpt + '-embedding', name), 'rb').read(), np.uint8)
    assert np.array_equal(got, z['case%d/file/%s' % (ci, name)]), name
  # 2. er_load_dense_embed (the native op's re-shard) and the tests' reader return what the reference's _load_embed returned
  loads = [k for k in z.files if k.startswith('case%d/load/' % ci)]
  assert loads
  for key in loads:
    r, w_new = [int(x) for x in key.rsplit('/', 1)[1].split('_of_')]
    want = z[key]
    n_new = (rows + w_new - 1) // w_new
    assert want.shape == (n_new, dim)
    assert np.array_equal(ref_load_embed(ckpt + '-embedding', FILE_VAR, dim, n_new, r, w_new), want), key
    assert np.array_equal(be.load_dense_embed(ckpt, FILE_VAR, r, w_new, dim, n_new), want), key


def test_pure_python_crc32c_known_answers():
  # RFC 3720 B.4 test vectors of CRC-32C
  assert crc32c(b'\x00' * 32) == 0x8A9136AA
  assert crc32c(b'\xff' * 32) == 0x62A8AB43
  assert crc32c(bytes(range(32))) == 0x46DD794E
  assert crc32c(b'123456789') == 0xE3069283


def test_tensor_bundle_files_through_an_independent_reader(tmp_path, built_lib):
  """utils/tensor_bundle.py's files read back by tests/_ckpt_readers.read_bundle: its own table parser, the real protobuf
  runtime for the protos, a pure-python CRC-32C.  Enough variables for several data blocks and restart intervals."""
  from easyrec_amd.utils import tensor_bundle
  rng = np.random.default_rng(5)
  tensors = {}
  for i in range(70):
    tensors['deep_feature/dnn_%d/kernel' % i] = rng.standard_normal((7 + i, 5)).astype(np.float32)
    tensors['deep_feature/dnn_%d/kernel/Adam' % i] = rng.standard_normal((7 + i, 5)).astype(np.float32)
  tensors['global_step'] = np.array(1234, dtype=np.int64)
  tensors['big/table'] = rng.standard_normal((300, 300)).astype(np.float32)  # > one 256 KiB data block of the data file
  tensors['empty'] = np.zeros((0, 4), dtype=np.float32)
  prefix = os.path.join(str(tmp_path), 'model.ckpt-9')
  tensor_bundle.write_bundle(prefix, tensors)
  got = read_bundle(prefix)
  assert set(got) == set(tensors)
  for k, v in tensors.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k


def test_checkpoint_state_keeps_its_history_and_is_replaced_atomically(tmp_path, built_lib):
  """tf.train.Saver appends to all_model_checkpoint_paths (oldest first); a state file that named only the newest prefix
  dropped the directory's history (round-4 advisor finding).  Entries whose files are gone are dropped."""
  from easyrec_amd.utils import tensor_bundle
  d = str(tmp_path)
  for step in (10, 20, 30):
    prefix = os.path.join(d, 'model.ckpt-%d' % step)
    tensor_bundle.write_bundle(prefix, {'global_step': np.array(step, dtype=np.int64)})
    tensor_bundle.write_checkpoint_state(prefix)
    assert not [f for f in os.listdir(d) if f.endswith('.tmp')]
  lines = open(os.path.join(d, 'checkpoint')).read().splitlines()
  assert lines == ['model_checkpoint_path: "model.ckpt-30"', 'all_model_checkpoint_paths: "model.ckpt-10"',
                   'all_model_checkpoint_paths: "model.ckpt-20"', 'all_model_checkpoint_paths: "model.ckpt-30"']
  os.remove(os.path.join(d, 'model.ckpt-10.index'))
  tensor_bundle.write_checkpoint_state(os.path.join(d, 'model.ckpt-30'))  # (saving a prefix again does not duplicate it)
  lines = open(os.path.join(d, 'checkpoint')).read().splitlines()
  assert lines == ['model_checkpoint_path: "model.ckpt-30"', 'all_model_checkpoint_paths: "model.ckpt-20"',
                   'all_model_checkpoint_paths: "model.ckpt-30"']
  assert int(read_bundle(os.path.join(d, 'model.ckpt-20'))['global_step']) == 20
