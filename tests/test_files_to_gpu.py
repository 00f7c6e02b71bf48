"""-m gpu: files -> reader -> HIP training step, against the model-level CPU oracle (a1 / config 1, f1, f2).

  * BASELINE config 1: a TSV file through `CSVInput` (input/csv_input.py:33-76) - native decode, id strings hashed ON
    THE DEVICE (er_hash_bucket_fast) - into `EasyRecEstimator.train_step` on cuda:0, eager and as a replayed hipGraph.
  * f1: `ParquetInput` packed `sparse_fea` / `dense_fea` batches (input/parquet_input.py:204-252) and `CriteoInput`
    binary batches (input/criteo_binary_reader.py) as ONE packed arena copy per step into the HIP path.
  * f2: `checkpoint.save` -> `restore` into a freshly built estimator -> continue, on cuda:0, dense tables and
    hash-table (`ev_params`) tables, Adam and lazy Adam (compat/embedding_parallel_saver.py:99-190, 187-280).
"""
import logging
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402
from oracle.model_oracle import OracleTrainer  # noqa: E402

logging.disable(logging.WARNING)
DEV = 'cuda:0'


def _cfg(name):
  return config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', name))


def _write_criteo_tsv(path, n, seed, vocab=200):
  """label \\t 13 integer features (some empty) \\t 26 eight-hex-digit ids (some empty; few distinct values, so that
  ids repeat inside a batch) - the Criteo format of examples/data/criteo."""
  rng = np.random.default_rng(seed)
  pool = ['%08x' % v for v in rng.integers(0, 2**32, size=vocab)]
  with open(path, 'w') as f:
    for i in range(n):
      ints = ['%d' % rng.integers(0, 60) if rng.random() > 0.15 else '' for _ in range(13)]
      cats = [pool[int(rng.integers(0, vocab))] if rng.random() > 0.1 else '' for _ in range(26)]
      f.write('\t'.join(['%d' % int(rng.random() < 0.3)] + ints + cats) + '\n')


def _close(got, exp, step):
  """north_star's bar on the first step from identical parameters (1e-4 on the losses); later steps within 2e-3: two
  fp32 implementations drift apart through Adam's normalisation of near-zero gradients (tests/test_deepfm_gpu.py)."""
  tol = 1e-4 if step == 0 else 2e-3
  for k in exp:
    assert abs(got[k] - exp[k]) <= tol * max(1.0, abs(exp[k])), (step, k, got[k], exp[k])


def _compare_slots(est, orc, tol=5e-4):
  """After the FIRST step: the gradients of every variable, read back as Adam's first moment, within tol of the tensor's
  scale (+ 1e-6 of the model's largest gradient scale for tensors whose true gradient is rounding noise)."""
  st = est.state_dict(slots=True)
  names = set(orc.state)
  n = 0
  gmax = max(float(np.max(np.abs(v))) for kk, v in orc.slots.items() if kk.endswith('/m'))
  for k in orc.state:
    key = k + '/m'
    if key not in orc.slots or key not in st:
      continue
    if k.endswith('/bias') and (k[:-len('/bias')] + '/bn/gamma') in names:
      continue  # d(loss)/d(bias) == 0 under BatchNorm: rounding noise on both sides
    ref = orc.slots[key]
    scale = float(np.max(np.abs(ref))) + 1e-30
    assert float(np.max(np.abs(st[key] - ref))) <= tol * scale + 1e-6 * gmax, (key, float(np.max(np.abs(st[key] - ref))), scale)
    n += 1
  assert n > 10


@pytest.mark.parametrize('graph', [False, True])
def test_csv_file_trains_on_the_gpu_like_the_oracle(tmp_path, graph):
  """BASELINE config 1 on the HIP path: TSV -> CSVInput -> (packed arena copy) -> device hashing -> DeepFM step."""
  from easyrec_amd.input.input import Input
  cfg = _cfg('deepfm_criteo_small.config')
  B, steps = 64, 4
  path = os.path.join(str(tmp_path), 'train.tsv')
  _write_criteo_tsv(path, B * steps + 7, seed=21)
  inp = Input.create(cfg.data_config, cfg.feature_config.features, path, batch_size=B, hash_on_host=False)
  assert type(inp).__name__ == 'CSVInput' and inp._native_ok()
  est = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=9).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  n = 0
  for b in inp.batches(num_epochs=1, drop_remainder=True):
    assert 'str_bytes' in b, 'the id strings must reach the device un-hashed'
    if graph and n == 1:
      est.capture(warmup=0)  # (after one eager step; capturing executes nothing, the next train_step replays it)
    est.train_step(est.features.pack(b))
    _close(est.loss_values(), orc.train_step(b), n)
    if n == 0:
      _compare_slots(est, orc)
    n += 1
  assert n == steps and (est.graph is not None) == graph


def test_csv_device_hash_equals_host_hash(tmp_path):
  """The same file read twice: ids hashed by the host reader (oracle-pinned FarmHash port) and by er_hash_bucket_fast on
  the device after the packed copy - the id buffers the lookups read must be identical."""
  from easyrec_amd.input.csv_input import CSVInput
  from easyrec_amd.input.features import DeviceFeatures
  cfg = _cfg('deepfm_criteo_small.config')
  B = 96
  path = os.path.join(str(tmp_path), 'rows.tsv')
  _write_criteo_tsv(path, B, seed=4, vocab=50)
  feats = list(cfg.feature_config.features)
  host = next(CSVInput(cfg.data_config, feats, path, batch_size=B, hash_on_host=True).batches())
  dev = next(CSVInput(cfg.data_config, feats, path, batch_size=B, hash_on_host=False).batches())
  from easyrec_amd.input.features import FeatureSchema
  schema = FeatureSchema(cfg.data_config, feats, batch_size=B)
  df = DeviceFeatures(schema, DEV)
  df.load(df.pack(dev))
  df.transform()
  assert np.array_equal(df.hash_ids.cpu().numpy(), np.asarray(host['hash_ids']))
  assert np.array_equal(df.raw_block.cpu().numpy(), np.asarray(host['raw'], dtype=np.float32))


def test_parquet_packed_batches_train_on_the_gpu_like_the_oracle(tmp_path):
  """f1: ParquetInput's packed batches (ids already integers: `vals % num_buckets`, ragged tag offsets, un-normalised
  dense values) drive the HIP step; losses and gradients follow the oracle on the same batches."""
  pytest.importorskip('pyarrow')
  import make_input_vectors as mv
  from easyrec_amd.input.input import Input
  cfg = _cfg('deepfm_parquet_small.config')
  paths = mv.write_parquet_files(str(tmp_path))
  reader = Input.create(cfg.data_config, cfg.feature_config.features, ','.join(paths))
  assert type(reader).__name__ == 'ParquetInput'
  B = cfg.data_config.batch_size
  est = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=2).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  n = 0
  for b in reader.batches():
    packed = est.features.pack(b)
    assert 'packed' in packed and packed['packed'].dtype == torch.uint8
    est.train_step(packed)
    _close(est.loss_values(), orc.train_step(b), n)
    if n == 0:
      _compare_slots(est, orc)
    n += 1
  assert n == 6


def test_criteo_binary_batches_train_on_the_gpu_like_the_oracle(tmp_path):
  """f1: CriteoInput (label / dense / category binary parts) -> device hashing of the stringified categories -> DeepFM
  step on the HIP path against the oracle."""
  import make_input_vectors as mv
  from easyrec_amd.input.criteo_input import CriteoInput
  cfg = _cfg('deepfm_criteo_small.config')
  lbl, dense, cat = mv.write_criteo_files(str(tmp_path))
  B = 16
  reader = CriteoInput(cfg.data_config, cfg.feature_config.features,
                       {'label_path': lbl[:1], 'dense_path': dense[:1], 'category_path': cat[:1]}, batch_size=B,
                       hash_on_host=False)
  est = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=6).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  n = 0
  for b in reader.batches(num_epochs=1):
    est.train_step(est.features.pack(b))
    _close(est.loss_values(), orc.train_step(b), n)
    if n == 0:
      _compare_slots(est, orc)
    n += 1
  assert n == 40 // B


def _train(est, batches):
  out = []
  for b in batches:
    est.train_step(b)
    out.append(est.loss_values())
  return out


@pytest.mark.parametrize('optimizer', [None, 'lazy'])
@pytest.mark.parametrize('graph', [False, True])
def test_save_restore_continues_bit_identically_on_the_gpu(tmp_path, optimizer, graph):
  """f2 on cuda:0: 3 steps, save (lazily decayed rows are flushed first), restore into a freshly built estimator with
  another seed, 2 more steps == 5 uninterrupted steps, bit for bit: losses, tables, slots, dense variables.  graph=True:
  the continuing estimators replay a captured hipGraph (restore happens before capture)."""
  from test_embedding_parallel import _cfg_and_batches
  from easyrec_amd.utils import checkpoint
  B = 16
  cfg, batches = _cfg_and_batches(B, 5, optimizer)
  a = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=3).build()
  _train(a, batches[:3])
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-3')
  checkpoint.save(a, ckpt)
  files = os.listdir(ckpt + '-embedding')
  assert 'embed-input_layer__C1_embedding__embedding_weights:0-part-0.bin' in files, files[:4]
  assert any(f.endswith('embedding_weights__Adam_1:0-part-0.bin') for f in files)
  b = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=99).build()
  if graph:
    _train(b, batches[:1])  # (one eager step before a capture: lazily created workspaces exist; the restore overwrites it)
  checkpoint.restore(b, ckpt)
  assert b.global_step == 3
  if graph:
    for e in (a, b):
      e.capture(warmup=0)
  rest_a = _train(a, batches[3:])
  rest_b = _train(b, batches[3:])
  assert rest_a == rest_b
  sa, sb = a.state_dict(slots=True), b.state_dict(slots=True)
  assert set(sa) == set(sb)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


def test_restore_into_a_trained_estimator_replays_no_stale_decay(tmp_path):
  """Restoring into an estimator that has ALREADY trained past the checkpoint (pending lazy decay on its rows, a longer
  lr_t history): the restored tables must not receive the old run's pending decay (set_global_step marks the rows
  current before anything can flush)."""
  from test_embedding_parallel import _cfg_and_batches
  from easyrec_amd.utils import checkpoint
  B = 16
  cfg, batches = _cfg_and_batches(B, 6, None)
  a = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=3).build()
  _train(a, batches[:2])
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-2')
  checkpoint.save(a, ckpt)
  want = _train(a, batches[2:4])
  b = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=3).build()
  _train(b, batches[:5])  # rows carry pending decay of steps this checkpoint never saw
  checkpoint.restore(b, ckpt)
  got = _train(b, batches[2:4])
  assert got == want


def test_hash_table_tables_save_restore_continue_on_the_gpu(tmp_path):
  """f2, `ev_params` tables on cuda:0: key / value files per table and slot, restored into another arena order."""
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.utils import checkpoint
  B = 16
  cfg = _cfg('deepfm_kv_criteo_small.config')
  gen = SyntheticCriteo(cfg.data_config, list(cfg.feature_config.features), batch_size=B, seed=5)
  batches = [gen.next_batch() for _ in range(5)]
  a = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=3).build()
  assert a.engine.kv_tables
  _train(a, batches[:3])
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-3')
  # stale parts of a larger previous world must disappear when worker 0 saves (embedding_parallel_saver.py:207-216)
  kv_name = next(iter(a.engine.kv_tables))
  stem = checkpoint.embed_file_var_name(kv_name)
  os.makedirs(ckpt + '-embedding')
  for ext in ('key', 'val'):
    open(os.path.join(ckpt + '-embedding', '%s-part-3.%s' % (stem, ext)), 'wb').write(b'\0' * 8)
  checkpoint.save(a, ckpt)
  files = os.listdir(ckpt + '-embedding')
  assert stem + '-part-0.key' in files and stem + '-part-0.val' in files, files[:6]
  assert stem + '-part-3.key' not in files and stem + '-part-3.val' not in files
  rest_a = _train(a, batches[3:])
  b = EasyRecEstimator(cfg, device=DEV, batch_size=B, seed=99).build()
  checkpoint.restore(b, ckpt)
  assert b.global_step == 3
  rest_b = _train(b, batches[3:])
  assert rest_a == rest_b
  sa, sb = a.state_dict(slots=True), b.state_dict(slots=True)
  assert set(sa) == set(sb)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


@pytest.mark.parametrize('optimizer', [None, 'lazy'])
def test_what_a_gpu_run_writes_reads_back_through_the_reference_format(tmp_path, optimizer):
  written_files_check(DEV, tmp_path, optimizer)


def written_files_check(device, tmp_path, optimizer):
  """f2, the other half: not save -> own restore, but the FILES a cuda:0 run writes against the reference's formats.
  Every `embed-<var>-part-0.bin` (tables and their Adam slots) is read with the reference's re-shard logic
  (`_load_embed`, compat/embedding_parallel_saver.py:140-168: tests/_ckpt_readers.ref_load_embed, held to the reference's
  own function by tests/test_checkpoint_pins.py) - as the 1 worker that wrote it and as worker r of 3 - and must hold
  exactly state_dict()'s rows; `<ckpt>.index` + `.data-00000-of-00001` are read by an independent tensor-bundle reader
  (own table parser, the protobuf runtime, a pure-python CRC-32C: model/easy_rec_model.py:219-351 reads them with
  tf.train.NewCheckpointReader) and must hold every dense variable, its Adam slots and global_step under their TF names."""
  from _ckpt_readers import read_bundle, ref_load_embed
  from test_embedding_parallel import _cfg_and_batches
  from easyrec_amd.utils import checkpoint
  B = 16
  cfg, batches = _cfg_and_batches(B, 3, optimizer)
  est = EasyRecEstimator(cfg, device=device, batch_size=B, seed=3).build()
  _train(est, batches)
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-3')
  checkpoint.save(est, ckpt)
  state = est.state_dict(slots=True)
  slot_names = {'m': 'Adam', 'v': 'Adam_1'}
  folder = ckpt + '-embedding'
  n_files = 0
  for name, t in est.engine.tables.items():
    for key, var in [(name, name)] + [(name + '/' + s, name + '/' + suffix) for s, suffix in slot_names.items()]:
      want = state[key]
      rows, dim = want.shape
      fv = 'embed-' + (var + ':0').replace('/', '__')
      path = os.path.join(folder, fv + '-part-0.bin')
      assert os.path.getsize(path) == rows * dim * 4, path
      n_files += 1
      assert np.array_equal(ref_load_embed(folder, fv, dim, rows, 0, 1), want), key
      n3 = (rows + 2) // 3
      for r in range(3):  # the same file re-sharded for a world of 3, as a reference worker would load it
        got = ref_load_embed(folder, fv, dim, n3, r, 3)
        mine = want[r::3]
        assert np.array_equal(got[:len(mine)], mine) and not got[len(mine):].any(), (key, r)
  assert n_files == 3 * len(est.engine.tables) and n_files == len(os.listdir(folder))
  bundle = read_bundle(ckpt)
  assert int(bundle['global_step']) == 3 and bundle['global_step'].dtype == np.int64
  dense = est.varstore.state_dict()
  assert len(dense) > 10
  for k, v in dense.items():
    assert np.array_equal(bundle[k], np.asarray(v)), k
  vs = est.varstore
  n_slots = 0
  for name in vs.trainable_names():
    o, n = vs._offsets[name]
    for s, suffix in slot_names.items():
      want = vs.slots[s][o:o + n].view(vs._vars[name]['tensor'].shape).cpu().numpy()
      assert np.array_equal(bundle[name + '/' + suffix], want), (name, suffix)
      n_slots += 1
  assert n_slots == 2 * len(vs.trainable_names()) and any(np.abs(bundle[k]).max() > 0 for k in bundle if k.endswith('/Adam_1'))
  # TF variable names of the reference's graph (layers/dnn.py:57-79 under the model's scopes), not names of this package
  assert any(k.endswith('/kernel') for k in bundle) and any(k.endswith('/bias') for k in bundle) and \
      any(k.endswith('/moving_mean') or k.endswith('/moving_variance') for k in bundle), sorted(bundle)[:8]
