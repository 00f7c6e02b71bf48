"""Checkpoints with the reference's sharded embedding files.

Reference: easy_rec/python/compat/embedding_parallel_saver.py:99-190 - under embedding parallelism every worker
writes its shard of every embedding variable to `<ckpt>-embedding/embed-<var name, '/' -> '__'>-part-<rank>.bin`
(raw float32 rows; row i of worker k is table row i * W + k) and on restore re-shards whatever parts it finds by
`row % world` (native op ops/src/load_dense_embed.cc -> er_load_dense_embed).  The optimizer's slot variables are
sharded the same way and use TF's slot names (`<var>/Adam`, `<var>/Adam_1`, `<var>/Adagrad`).

Hash-table (`ev_params`) tables are written the way the saver writes SOK dynamic variables (:187-222): the ids that have
a row as `embed-<var>-part-<rank>.key` (int64) and their rows as `...-part-<rank>.val` (float32 [n, dim]) - never the
arena, whose row order is an accident of the run - and restored through er_load_kv_embed (ops/src/load_kv_embed.cc:
115-163: keys with `key % world == rank`), re-inserted into the map (any arena order), values and slots scattered to
the rows they got.

Everything else - dense variables, their slots, the step - goes through the TF Saver in the reference (model/
easy_rec_model.py:219-351 restores them by variable name): tensor-bundle files `<ckpt>.index` + `<ckpt>.data-00000-of-00001`
under the same TF variable names (`deep_feature/dnn_0/kernel`, `.../kernel/Adam`, `global_step`), written and read by
utils/tensor_bundle.py, plus the `checkpoint` state file.  (Checkpoints of earlier rounds - `<ckpt>.dense.npz` - still load.)

A checkpoint written by W workers loads on any other number (1 GPU <-> 8 GPUs): that is what the re-shard is for.
"""
import os

import numpy as np
import torch

from easyrec_amd import kernels
from easyrec_amd.utils import tensor_bundle

_SLOT_NAMES = {  # estimator slot -> TF slot variable suffix, per optimizer kind
    kernels.OPT_ADAM: {'m': 'Adam', 'v': 'Adam_1'},
    kernels.OPT_LAZY_ADAM: {'m': 'Adam', 'v': 'Adam_1'},
    kernels.OPT_ADAGRAD: {'v': 'Adagrad'},
    kernels.OPT_SGD: {},
}


def embed_file_var_name(tf_var_name):
  """embedding_parallel_saver.py:103-111: 'embed-' + embed_var.name with '/' -> '__' (TF names end in ':0')."""
  return 'embed-' + (tf_var_name + ':0').replace('/', '__')


def _host_backend():
  # file I/O entry points of libeasyrec_hip.so: usable without a GPU, whatever backend the layers run on
  return kernels.HipBackend()


def _engine_tables(engine):
  """[(table name, sharded?, local rows tensor getter)]"""
  sharded = hasattr(engine, 'placement')
  for name in engine.tables:
    is_shard = sharded and engine.placement[name][0] != 'rep'
    yield name, is_shard


def _kv_file(ckpt_path, var_name, rank, ext):
  return '%s-embedding/%s-part-%d.%s' % (ckpt_path, var_name, rank, ext)


def _remove_stale_kv_parts(ckpt_path, var_name, world):
  """embedding_parallel_saver.py:207-216: worker 0 deletes `part-<id>.key/.val` of workers that no longer exist (id >=
  world) - a loader reads EVERY part it finds, so parts left by a larger previous world would come back as stale or
  duplicate keys."""
  import glob
  import re
  for ext in ('key', 'val', 'seen', 'freq', 'version'):
    for path in glob.glob(glob.escape('%s-embedding/%s-part-' % (ckpt_path, var_name)) + '*.' + ext):
      m = re.search(r'-part-(\d+)\.%s$' % ext, path)
      if m and int(m.group(1)) >= world:
        os.remove(path)


def _save_kv_table(engine, ckpt_path, name, slots, rank, world=1):
  """keys / rows of the ids that have a row, and the same rows of every slot (the slots share the table's keys).  An
  `ev_params { filter_freq / steps_to_live }` table also writes its filter's state next to them: `.seen` (every id the
  table tracks, int64), `.freq` (its count, int32) and `.version` (the step of its last lookup, int32)."""
  kv = engine.kv_tables[name]
  os.makedirs(ckpt_path + '-embedding', exist_ok=True)
  fv0 = embed_file_var_name(name)
  if engine._kv_filtered(name):
    seen, seen_rows, freq, version = kernels.hip().kv_export_all(kv)
    has_row = seen_rows >= 0
    keys, rows = seen[has_row], seen_rows[has_row]
    seen.cpu().numpy().astype(np.int64).tofile(_kv_file(ckpt_path, fv0, rank, 'seen'))
    np.minimum(freq.cpu().numpy(), max(engine.tables[name]['kv_filter_freq'], 1)).astype(np.int32).tofile(
        _kv_file(ckpt_path, fv0, rank, 'freq'))
    version.cpu().numpy().astype(np.int32).tofile(_kv_file(ckpt_path, fv0, rank, 'version'))
  else:
    keys, rows = kernels.hip().kv_export(kv)
  keys_np = keys.cpu().numpy().astype(np.int64)
  pairs = [(name, engine.table_view(name))] + [(name + '/' + suffix, engine.slot_view(name, s)) for s, suffix in slots.items()]
  for var, view in pairs:
    if view is None:
      continue
    fv = embed_file_var_name(var)
    keys_np.tofile(_kv_file(ckpt_path, fv, rank, 'key'))
    view[rows.to(view.device)].detach().cpu().numpy().astype(np.float32).tofile(_kv_file(ckpt_path, fv, rank, 'val'))
    if rank == 0:
      _remove_stale_kv_parts(ckpt_path, fv, world)


def _load_kv_filter_parts(ckpt_path, var_name, rank, world):
  """The filter's state of the ids this rank owns (id % world == rank) out of every part found, ascending; None when
  the checkpoint has none."""
  import glob
  seen, freq, version = [], [], []
  for path in sorted(glob.glob(glob.escape('%s-embedding/%s-part-' % (ckpt_path, var_name)) + '*.seen')):
    k = np.fromfile(path, dtype=np.int64)
    mine = (k % world) == rank
    seen.append(k[mine])
    freq.append(np.fromfile(path[:-len('seen')] + 'freq', dtype=np.int32)[mine])
    version.append(np.fromfile(path[:-len('seen')] + 'version', dtype=np.int32)[mine])
  if not seen:
    return None, None, None
  seen, freq, version = np.concatenate(seen), np.concatenate(freq), np.concatenate(version)
  order = np.argsort(seen, kind='stable')
  return seen[order], freq[order], version[order]


def _restore_kv_table(be, engine, ckpt_path, name, slots, rank, world):
  """The table becomes the saved one (layers/input_layer.py load_kv_table): this rank's ids out of every part (the
  loader keeps id % world == rank, ops/src/load_kv_embed.cc:100-130); a slot's file carries the table's keys in its own
  order."""
  dim = engine.table_view(name).shape[1]
  keys, vals = be.load_kv_embed(ckpt_path, embed_file_var_name(name), rank, world, dim)
  keys = np.ascontiguousarray(keys, dtype=np.int64)
  order = np.argsort(keys, kind='stable')
  keys, vals = keys[order], np.ascontiguousarray(vals, dtype=np.float32)[order]
  slot_values = {}
  for s, suffix in slots.items():
    if engine.slot_view(name, s) is None:
      continue
    skeys, svals = be.load_kv_embed(ckpt_path, embed_file_var_name(name + '/' + suffix), rank, world, dim)
    so = np.argsort(np.asarray(skeys, dtype=np.int64), kind='stable')
    assert np.array_equal(np.asarray(skeys, dtype=np.int64)[so], keys), \
        'checkpoint %s: slot file %s does not hold the table file\'s ids' % (ckpt_path, name + '/' + suffix)
    slot_values[s] = np.ascontiguousarray(svals, dtype=np.float32)[so]
  seen, freq, version = _load_kv_filter_parts(ckpt_path, embed_file_var_name(name), rank, world)
  # (EmbeddingEngine.load_kv_table, not the sharded override: the ids are already this rank's)
  from easyrec_amd.layers.input_layer import EmbeddingEngine
  EmbeddingEngine.load_kv_table(engine, name, keys, vals, slot_values, seen, freq, version)


def save(est, ckpt_path):
  """Every rank calls it.  Rank 0 writes the dense file; every rank its embedding shards (replicated and
  single-GPU tables: rank 0 alone, as a one-part table)."""
  be = _host_backend()
  engine = est.engine
  rank, world = getattr(engine, 'rank', 0), getattr(engine, 'world', 1)
  engine.flush_decay()
  if est.device.type == 'cuda':
    torch.cuda.synchronize()
  # an overflowed fixed-capacity exchange or hash-table arena voids the steps since: never persist tables it may have
  # touched - and never EXPORT an overflowed map (evict_stale -> er_kv_export_all sizes its buffers for a map within bounds)
  if hasattr(engine, 'check_overflow'):
    engine.check_overflow()
  if hasattr(engine, 'evict_stale'):
    engine.evict_stale(est.global_step)  # ev_params.steps_to_live: eviction happens when a checkpoint is written
  os.makedirs(os.path.dirname(os.path.abspath(ckpt_path)) or '.', exist_ok=True)
  slots = _SLOT_NAMES[est.opt_emb.kind]
  for name, is_shard in _engine_tables(engine):
    if not is_shard and rank != 0:
      continue
    t_idx, t_num = (rank, world) if is_shard else (0, 1)
    if engine.tables[name].get('kv'):
      _save_kv_table(engine, ckpt_path, name, slots, t_idx, t_num)
      continue
    be.save_dense_embed(ckpt_path, embed_file_var_name(name), t_idx, t_num, engine.table_view(name).cpu().numpy())
    for s, suffix in slots.items():
      sv = engine.slot_view(name, s)
      if sv is not None:
        be.save_dense_embed(ckpt_path, embed_file_var_name(name + '/' + suffix), t_idx, t_num, sv.cpu().numpy())
  if rank == 0:
    dense = {}
    vs = est.varstore
    for k, v in vs.state_dict().items():
      dense[k] = np.asarray(v)
    dslots = _SLOT_NAMES[est.opt_dense.kind]
    for name in vs.trainable_names():
      o, n = vs._offsets[name]
      for s, suffix in dslots.items():
        if s in vs.slots:
          dense[name + '/' + suffix] = vs.slots[s][o:o + n].view(vs._vars[name]['tensor'].shape).cpu().numpy().copy()
    # hash-table tables: the generator of rows created AFTER the restore (seed, mean, stddev), so that a resumed run draws
    # the rows an uninterrupted one would
    for name, kv in getattr(engine, 'kv_tables', {}).items():
      dense[name + '/kv_meta'] = np.array([kv['seed'], kv['mean'], kv['stddev'], kv['capacity']], dtype=np.float64)
      dense[name + '/kv_seed'] = np.asarray(int(kv['seed']), dtype=np.int64)  # (float64 loses seeds >= 2^53)
    dense['global_step'] = np.asarray(int(est.global_step), dtype=np.int64)
    tensor_bundle.write_bundle(ckpt_path, dense)
    tensor_bundle.write_checkpoint_state(ckpt_path)


def restore(est, ckpt_path):
  """Every rank calls it; the embedding parts may have been written by a different number of workers."""
  be = _host_backend()
  engine = est.engine
  rank, world = getattr(engine, 'rank', 0), getattr(engine, 'world', 1)
  slots = _SLOT_NAMES[est.opt_emb.kind]
  class _Arrays(dict):  # (np.load's interface: .files + item access)
    files = property(lambda self: list(self.keys()))
  if os.path.exists(ckpt_path + '.index'):
    z = _Arrays(tensor_bundle.read_bundle(ckpt_path))
  else:  # a checkpoint of rounds 2-3
    z0 = np.load(ckpt_path + '.dense.npz')
    z = _Arrays((k, z0[k]) for k in z0.files)
  for name, is_shard in _engine_tables(engine):
    t_idx, t_num = (rank, world) if is_shard else (0, 1)
    if engine.tables[name].get('kv'):
      _restore_kv_table(be, engine, ckpt_path, name, slots, t_idx, t_num)
      continue
    view = engine.table_view(name)
    n_local, dim = view.shape
    view.copy_(torch.from_numpy(be.load_dense_embed(ckpt_path, embed_file_var_name(name), t_idx, t_num, dim, n_local)))
    for s, suffix in slots.items():
      sv = engine.slot_view(name, s)
      if sv is not None:
        sv.copy_(torch.from_numpy(
            be.load_dense_embed(ckpt_path, embed_file_var_name(name + '/' + suffix), t_idx, t_num, dim, n_local)))
  for name, kv in getattr(engine, 'kv_tables', {}).items():
    if name + '/kv_meta' in z.files:
      meta = z[name + '/kv_meta']
      kv['seed'], kv['mean'], kv['stddev'] = int(meta[0]), float(meta[1]), float(meta[2])
      if name + '/kv_seed' in z.files:
        kv['seed'] = int(z[name + '/kv_seed'])
      engine._kv_handle = None  # (the device-resident job table carries the generator's parameters: rebuilt on next use)
  vs = est.varstore
  vs.load_state_dict({k: z[k] for k in z.files}, strict=False)
  dslots = _SLOT_NAMES[est.opt_dense.kind]
  for name in vs.trainable_names():
    o, n = vs._offsets[name]
    for s, suffix in dslots.items():
      key = name + '/' + suffix
      if s in vs.slots and key in z.files:
        vs.slots[s][o:o + n].copy_(torch.from_numpy(z[key].reshape(-1)).to(vs.slots[s].device))
  est.set_global_step(int(z['global_step']))
