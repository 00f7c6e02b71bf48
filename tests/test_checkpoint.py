"""Sharded embedding checkpoints: the file format and re-sharding of
compat/embedding_parallel_saver.py + ops/src/load_dense_embed.cc / load_kv_embed.cc."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config')


def _ref_load_embed(folder, var_name, embed_dim, embed_part_size, part_id, part_num):
  """Restatement of the reference's python fallback `_load_embed` (embedding_parallel_saver.py:140-168)."""
  import glob
  files = glob.glob(os.path.join(folder, var_name + '-part-*.bin'))
  files.sort(key=lambda p: int(p.split('-')[-1].replace('.bin', '')))
  out = np.zeros([embed_part_size, embed_dim], dtype=np.float32)
  for f in files:
    part_id_o = int(f.split('-')[-1].replace('.bin', ''))
    val = np.frombuffer(open(f, 'rb').read(), np.float32).reshape([-1, embed_dim])
    ids_o = part_id_o + np.arange(len(val)) * len(files)
    sel = np.where(np.logical_and((ids_o % part_num) == part_id, ids_o < embed_part_size * part_num))[0]
    out[np.array(ids_o[sel] / part_num, dtype=np.int64)] = val[sel]
  return out


@pytest.mark.parametrize('rows,dim,w_old,w_new', [(103, 8, 3, 2), (64, 1, 1, 4), (1000, 16, 8, 1), (17, 4, 2, 5)])
def test_dense_embed_files_reshard_like_the_reference(tmp_path, built_lib, rows, dim, w_old, w_new):
  from easyrec_amd import kernels
  be = kernels.HipBackend()
  rng = np.random.default_rng(rows)
  table = rng.standard_normal((rows, dim)).astype(np.float32)
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-7')
  var = 'embed-input_layer__c1_embedding__embedding_weights:0'
  n_old = (rows + w_old - 1) // w_old
  # a stale part of a larger previous world must disappear when worker 0 saves
  os.makedirs(ckpt + '-embedding')
  open(os.path.join(ckpt + '-embedding', var + '-part-%d.bin' % (w_old + 2)), 'wb').write(b'junk')
  for k in range(w_old - 1, -1, -1):
    shard = np.zeros((n_old, dim), dtype=np.float32)
    mine = table[k::w_old]
    shard[:len(mine)] = mine
    be.save_dense_embed(ckpt, var, k, w_old, shard)
  names = sorted(os.listdir(ckpt + '-embedding'))
  assert names == sorted(var + '-part-%d.bin' % k for k in range(w_old))
  n_new = (rows + w_new - 1) // w_new
  full = np.zeros((n_new * w_new, dim), dtype=np.float32)
  for r in range(w_new):
    got = be.load_dense_embed(ckpt, var, r, w_new, dim, n_new)
    assert np.array_equal(got, _ref_load_embed(ckpt + '-embedding', var, dim, n_new, r, w_new))
    full[r::w_new] = got
  assert np.array_equal(full[:rows], table) and not full[rows:].any()
  with pytest.raises(RuntimeError):  # a shard size the files cannot fill: the reference op's consistency check
    be.load_dense_embed(ckpt, var, 0, w_new, dim, n_new + 5)


def test_kv_embed_files(tmp_path, built_lib):
  from easyrec_amd import kernels
  be = kernels.HipBackend()
  rng = np.random.default_rng(0)
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-1')
  os.makedirs(ckpt + '-embedding')
  var, dim = 'embed-kv_table:0', 4
  all_k, all_v = [], []
  for part in range(3):
    k = rng.integers(-10**12, 10**12, size=50 + part).astype(np.int64)
    v = rng.standard_normal((len(k), dim)).astype(np.float32)
    k.tofile(os.path.join(ckpt + '-embedding', '%s-part-%d.key' % (var, part)))
    v.tofile(os.path.join(ckpt + '-embedding', '%s-part-%d.val' % (var, part)))
    all_k.append(k)
    all_v.append(v)
  all_k, all_v = np.concatenate(all_k), np.concatenate(all_v)
  seen = 0
  for r in range(4):
    keys, vals = be.load_kv_embed(ckpt, var, r, 4, dim)
    sel = (all_k % 4) == r  # numpy's % is non-negative like the op's fixed-up remainder (load_kv_embed.cc:123-127)
    assert np.array_equal(keys, all_k[sel]) and np.array_equal(vals, all_v[sel])
    seen += len(keys)
  assert seen == len(all_k)


def _train(est, batches):
  out = []
  for b in batches:
    est.train_step(b)
    out.append(est.loss_values())
  return out


@pytest.mark.parametrize('optimizer', [None, 'lazy'])
def test_save_restore_continues_bit_identically(ref_backend, built_lib, tmp_path, optimizer):
  """3 steps, save, restore into a freshly built estimator (other seed), 2 more steps == 5 uninterrupted steps."""
  from test_embedding_parallel import _cfg_and_batches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import checkpoint
  B = 16
  cfg, batches = _cfg_and_batches(B, 5, optimizer)
  a = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  _train(a, batches[:3])
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-3')
  checkpoint.save(a, ckpt)
  files = os.listdir(ckpt + '-embedding')
  assert 'embed-input_layer__C1_embedding__embedding_weights:0-part-0.bin' in files, files[:4]
  assert any(f.endswith('embedding_weights__Adam_1:0-part-0.bin') for f in files)
  rest_a = _train(a, batches[3:])
  b = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=99).build()
  checkpoint.restore(b, ckpt)
  assert b.global_step == 3
  rest_b = _train(b, batches[3:])
  assert rest_a == rest_b
  sa, sb = a.state_dict(slots=True), b.state_dict(slots=True)
  assert set(sa) == set(sb)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


def test_hash_table_tables_save_restore_continue(ref_backend, built_lib, tmp_path):
  """`ev_params` tables: 3 steps, save (key / value files per table and slot, as the saver writes SOK variables:
  compat/embedding_parallel_saver.py:187-222), restore into a freshly built estimator with another seed (the ids get
  OTHER arena rows), 2 more steps == 5 uninterrupted steps: losses, and every id's row and slots."""
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import checkpoint, config_util
  B = 16
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_kv_criteo_small.config'))
  gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=5)
  batches = [gen.next_batch() for _ in range(5)]
  a = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  assert a.engine.kv_tables
  _train(a, batches[:3])
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-3')
  checkpoint.save(a, ckpt)
  files = os.listdir(ckpt + '-embedding')
  kv_name = next(iter(a.engine.kv_tables))
  stem = checkpoint.embed_file_var_name(kv_name)
  assert stem + '-part-0.key' in files and stem + '-part-0.val' in files, files[:6]
  assert checkpoint.embed_file_var_name(kv_name + '/Adam_1') + '-part-0.val' in files
  assert not any(f.startswith(stem) and f.endswith('.bin') for f in files), 'the arena itself must not be written'
  n_keys = os.path.getsize(os.path.join(ckpt + '-embedding', stem + '-part-0.key')) // 8
  assert 0 < n_keys <= 3 * B
  rest_a = _train(a, batches[3:])
  b = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=99).build()
  # a few ids before the restore: the restored ids land on other arena rows than in `a`
  checkpoint.restore(b, ckpt)
  assert b.global_step == 3
  rest_b = _train(b, batches[3:])
  assert rest_a == rest_b
  sa, sb = a.state_dict(slots=True), b.state_dict(slots=True)
  assert set(sa) == set(sb)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


def test_full_hash_table_arena_is_noticed_and_claims_no_slots(ref_backend):
  """More distinct ids than ev_params.max_capacity: new ids read zeros, the sticky flag raises at the estimator's
  periodic check / evaluate / state_dict, and the ids that found no row do not occupy map slots."""
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import config_util
  B = 16
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_kv_criteo_small.config'))
  for f in cfg.feature_config.features:
    if f.HasField('ev_params'):
      f.ev_params.max_capacity = 8
  if cfg.model_config.HasField('ev_params'):
    cfg.model_config.ev_params.max_capacity = 8
  gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=5, mode='uniform')
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  est.OVERFLOW_CHECK_EVERY = 2
  est.train_step(gen.next_batch())
  with pytest.raises(RuntimeError, match='max_capacity'):
    est.train_step(gen.next_batch())
  with pytest.raises(RuntimeError, match='max_capacity'):
    est.evaluate([gen.next_batch()])
  kv = next(iter(est.engine.kv_tables.values()))
  assert len(kv['map']) <= kv['capacity']


def _gloo_ckpt_worker(rank, world, port, B, out_dir):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(1)
  from easyrec_amd import kernels
  from oracle.kernel_ref import RefBackend
  kernels._BACKEND = RefBackend()
  from test_embedding_parallel import _cfg_and_batches
  from easyrec_amd.model.embedding_parallel import EmbeddingParallelEstimator
  from easyrec_amd.utils import checkpoint
  cfg, batches = _cfg_and_batches(B, 2)
  est = EmbeddingParallelEstimator(cfg, device='cpu', batch_size=B, seed=3, rank=rank, world=world,
                                   replicate_bytes=1024).build()
  for b in batches:
    est.train_step(b)
  checkpoint.save(est, os.path.join(out_dir, 'model.ckpt-2'))
  state = est.state_dict(slots=True)
  if rank == 0:
    np.savez(os.path.join(out_dir, 'state.npz'), **{k.replace('/', '|'): v for k, v in state.items()})
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_checkpoint_restores_on_one_rank(ref_backend, built_lib, tmp_path):
  """World 2 (gloo) trains and saves part-0 / part-1 files; a single-GPU estimator restores them: every table,
  slot and dense variable equals the two-rank run's gathered state (the re-shard by row % world)."""
  import torch.multiprocessing as mp
  from test_embedding_parallel import _cfg_and_batches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import checkpoint
  B, world = 24, 2
  port = 31500 + (os.getpid() % 2000)
  mp.spawn(_gloo_ckpt_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
  names = os.listdir(os.path.join(str(tmp_path), 'model.ckpt-2-embedding'))
  assert any(n.endswith('-part-1.bin') for n in names) and any(n.endswith('-part-0.bin') for n in names)
  cfg, _ = _cfg_and_batches(B, 2)
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=42).build()
  checkpoint.restore(est, os.path.join(str(tmp_path), 'model.ckpt-2'))
  want = {k.replace('|', '/'): v for k, v in np.load(os.path.join(str(tmp_path), 'state.npz')).items()}
  got = est.state_dict(slots=True)
  assert set(got) == set(want)
  for k in want:
    assert np.array_equal(got[k], want[k]), k


def test_tensor_bundle_files(tmp_path, built_lib):
  """utils/tensor_bundle.py: the dense variables' checkpoint in TensorFlow's tensor-bundle format.  Writer against reader;
  the format's invariants a TF reader relies on (sorted unique keys with the header under "", footer magic, per-block and
  per-tensor masked CRC-32C, BlockHandles); the CRC against its published known answers (RFC 3720 B.4); a table of more than
  one data block; corruption is detected."""
  import struct
  from easyrec_amd.utils import tensor_bundle as tb
  assert tb._crc32c(b'123456789') == 0xE3069283 and tb._crc32c(bytes(32)) == 0x8A9136AA and tb._crc32c(b'\xff' * 32) == 0x62A8AB43
  assert tb._unmask(tb._mask(0x12345678)) == 0x12345678 and tb._mask(0) == 0xa282ead8
  rng = np.random.default_rng(0)
  tensors = {'deep_feature/dnn_0/kernel': rng.standard_normal((37, 16)).astype(np.float32),
             'deep_feature/dnn_0/kernel/Adam': rng.standard_normal((37, 16)).astype(np.float32),
             'deep_feature/dnn_0/bias': np.zeros(16, dtype=np.float32), 'global_step': np.asarray(12345, dtype=np.int64),
             'empty': np.zeros((0, 4), dtype=np.float32), 'a/double': rng.standard_normal(3)}
  for i in range(6000):  # enough index entries for more than one 256 KB data block
    tensors['many/var_%05d/with_a_long_common_prefix_in_its_name_%s' % (i, 'x' * 60)] = np.asarray([i], dtype=np.int32)
  prefix = os.path.join(str(tmp_path), 'model.ckpt-12345')
  tb.write_bundle(prefix, tensors)
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 2 * 256 * 1024  # (more than two data blocks)
  items = tb._read_table(prefix + '.index')
  keys = [k for k, _ in items]
  assert keys[0] == b'' and keys == sorted(keys) and len(set(keys)) == len(keys) == len(tensors) + 1
  assert items[0][1] == b'\x08\x01\x1a\x02\x08\x01'  # num_shards 1, (endianness LITTLE = default), version{producer 1}
  assert os.path.getsize(tb.data_file(prefix)) == sum(v.nbytes for v in tensors.values())
  got = tb.read_bundle(prefix)
  assert set(got) == set(tensors)
  for k, v in tensors.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
  # a flipped data byte fails the tensor's CRC; a flipped index byte fails a block's
  data = bytearray(open(tb.data_file(prefix), 'rb').read())
  data[100] ^= 1
  open(tb.data_file(prefix), 'wb').write(bytes(data))
  with pytest.raises(ValueError, match='checksum'):
    tb.read_bundle(prefix)
  idx = bytearray(raw)
  idx[1000] ^= 1
  open(prefix + '.index', 'wb').write(bytes(idx))
  with pytest.raises(ValueError, match='checksum'):
    tb._read_table(prefix + '.index')


@pytest.mark.parametrize('optimizer', [None, 'lazy'])
def test_written_files_read_back_through_the_reference_format_on_the_stand_in_backend(ref_backend, tmp_path, optimizer):
  """the -m gpu test of tests/test_files_to_gpu.py (files a run writes against the reference's formats), host logic here"""
  from test_files_to_gpu import written_files_check
  written_files_check('cpu', tmp_path, optimizer)
