"""Hash-table (`ev_params`) embedding tables: rows exist only for the ids seen in training, a new
row is a pure function of (seed, id, column), unseen ids read zeros at evaluation.  The product on the oracle's stand-in
backend against the oracle (which keeps a dict + arena of its own and assigns arena rows in its own order): losses,
the set of materialised ids, and every row by id."""
import os

import numpy as np

import _bars
import pytest

from easyrec_amd.utils import config_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(device, steps=3, B=64, config='deepfm_kv_criteo_small.config', n_kv=8, row_tol=2e-4, tie_rows=0, data_seed=12):
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
  est = EasyRecEstimator(cfg, device=device, batch_size=B, seed=4).build()
  kv_names = sorted(est.engine.kv_tables)
  assert len(kv_names) == n_kv, kv_names  # (deepfm: C1..C4, deep and wide)
  if 'criteo' not in config:
    SyntheticCriteo = SyntheticBatches  # noqa: N806  (schema-driven generator: tag lists)
  state0 = est.state_dict()
  assert all(state0[n + '/keys'].size == 0 and state0[n].shape[0] == 0 for n in kv_names)
  orc = OracleTrainer(cfg, state0, batch_size=B)
  gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=B, seed=data_seed)
  batches = [gen.next_batch() for _ in range(steps)]
  for step, b in enumerate(batches):
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(got[k] - exp[k]) <= 1e-4 * max(1.0, abs(exp[k])), (step, k, got[k], exp[k])
  st = est.state_dict(slots=True)
  for n in kv_names:
    keys, rows = orc.kv_state(n)
    assert np.array_equal(st[n + '/keys'], keys), n
    assert keys.size > 10 and st[n].shape == rows.shape
    # (after a few Adam steps: 2e-4 of the table's scale - elements whose gradient is rounding noise move by O(lr))
    _, m_rows = orc.kv_state(n, orc.slots[n + '/m'])
    m_scale = float(np.abs(m_rows).max())
    bad_m = np.abs(st[n + '/m'] - m_rows).max(axis=1) > row_tol * m_scale + 1e-9
    assert int(bad_m.sum()) <= tie_rows, (n, int(bad_m.sum()))
    # The rows themselves: Adam divides by sqrt(v), so a row whose gradient is small against the table's largest is
    # ill-conditioned - a difference far below the gradient tolerance above becomes an O(lr) difference of the row.
    # tie_rows == 0 (DeepFM): every row must agree.  tie_rows > 0 (MMoE on the GPU: the experts' pre-activations are
    # not normalised - their BatchNorm runs on the moving statistics, as the reference's - and the gates scale some
    # rows' gradients down to rounding level): the well-conditioned rows must agree but for `tie_rows` of them (a ReLU
    # input within rounding of zero flips with the GEMM's summation order), every row stays within a few lr-sized steps.
    diff = np.abs(st[n] - rows).max(axis=1)
    bad = diff > row_tol * float(np.abs(rows).max()) + 1e-7
    if tie_rows == 0:
      assert not bad.any(), (n, int(bad.sum()), float(diff.max()))
    else:
      strong = np.abs(m_rows).max(axis=1) >= 0.05 * m_scale
      assert int(strong.sum()) >= 5, (n, int(strong.sum()))
      assert int((bad & strong).sum()) <= tie_rows, (n, int((bad & strong).sum()), int(strong.sum()))
      assert float(diff.max()) <= 4e-3 * steps, n
  return est, cfg, batches, st


def _evaluate_and_reload(est, cfg, batches, st, device):
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  # evaluation creates no rows: ids never seen in training read zeros
  n_before = {n: st[n + '/keys'].size for n in est.engine.kv_tables}
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  fresh = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=est.batch_size, seed=99).next_batch()
  est.evaluate([fresh])
  after = est.state_dict()
  assert all(after[n + '/keys'].size == n_before[n] for n in n_before)
  # a restored estimator continues exactly like the original
  twin = EasyRecEstimator(cfg, device=device, batch_size=est.batch_size, seed=4).build()
  twin.load_state_dict(est.state_dict(slots=True))
  est.train_step(batches[0])
  twin.train_step(batches[0])
  a, b = est.loss_values(), twin.loss_values()
  assert all(abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(a[k])) for k in a), (a, b)


def test_kv_embeddings_match_the_oracle_on_the_stand_in_backend(ref_backend):
  est, cfg, batches, st = _run('cpu')
  _evaluate_and_reload(est, cfg, batches, st, 'cpu')


def test_kv_tag_features_match_the_oracle_on_the_stand_in_backend(ref_backend):
  """MMoE over the Taobao schema with hash-table backed IdFeatures AND TagFeatures (ragged id lists in fixed-capacity
  buffers: only the step's ids are translated)."""
  # (row tolerance 2e-3 of the table's scale: MMoE's gates make some row gradients rounding noise, which Adam turns
  # into O(lr) moves - the same allowance as the MMoE model tests)
  _run('cpu', config='mmoe_kv_taobao_small.config', n_kv=4, row_tol=2e-3)


def test_row_initialiser_is_a_pure_function_of_its_key():
  from oracle.kernel_ref import RefBackend
  a = RefBackend.kv_init_value(5, [3, 10**15, 7], 4, 0.0, 0.5)
  b = RefBackend.kv_init_value(5, [7, 3], 4, 0.0, 0.5)
  assert np.array_equal(a[0], b[1]) and np.array_equal(a[2], b[0])
  big = RefBackend.kv_init_value(1, np.arange(50000), 4, 0.25, 2.0)
  assert abs(float(big.mean()) - 0.25) < 0.02 and abs(float(big.std()) - 2.0) < 0.02
  assert float(np.abs(big - 0.25).max()) <= 2.0 * 3.4642  # the sum of four uniforms has bounded support


def _filtered_config():
  """deepfm_kv_criteo_small with ev_params { filter_freq / steps_to_live } on the hash-table features: C1 counter filter,
  C2 eviction, C3 both, C4 plain (feature_config.proto:27-29)."""
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_kv_criteo_small.config'))
  feats = cfg.feature_configs if cfg.feature_configs else cfg.feature_config.features
  by_name = {f.input_names[0]: f for f in feats}
  by_name['C1'].ev_params.filter_freq = 2
  by_name['C2'].ev_params.steps_to_live = 2
  by_name['C3'].ev_params.filter_freq = 3
  by_name['C3'].ev_params.steps_to_live = 3
  return cfg


def _compare_kv(est, orc, step, row_tol=2e-4, values=True):
  """The discrete state (which ids have a row, the counts, the stamps) always; the rows' values when `values`."""
  st = est.state_dict(slots=True)
  for n in sorted(est.engine.kv_tables):
    keys, rows = orc.kv_state(n)
    assert np.array_equal(st[n + '/keys'], keys), (step, n, st[n + '/keys'].size, keys.size)
    if keys.size and values:
      # Adam's first moment is well conditioned; the rows themselves are for the rows whose gradient is not rounding
      # noise (Adam turns a noise-level difference of m / sqrt(v) into an O(lr) difference: see _run above)
      _, m_rows = orc.kv_state(n, orc.slots[n + '/m'])
      m_scale = float(np.abs(m_rows).max())
      assert float(np.abs(st[n + '/m'] - m_rows).max()) <= row_tol * m_scale + 1e-9, (step, n)
      diff = np.abs(st[n] - rows).max(axis=1)
      strong = np.abs(m_rows).max(axis=1) >= 0.05 * m_scale
      assert float(diff[strong].max(initial=0.0)) <= row_tol * float(np.abs(rows).max()) + 1e-5, (step, n, float(diff[strong].max()))
      assert float(diff.max()) <= 4e-3, (step, n, float(diff.max()))
    if (n + '/kv_seen_keys') in st:
      seen, freq, version = orc.kv_filter_state(n)
      assert np.array_equal(st[n + '/kv_seen_keys'], seen), (step, n)
      assert np.array_equal(st[n + '/kv_freq'], freq), (step, n)
      assert np.array_equal(st[n + '/kv_version'], version), (step, n)
  return st


def _run_filtered(device, tmp_path, steps=5, B=64, data_seed=13, value_steps=100):
  """Counter filter + steps_to_live against the oracle: losses, the admitted ids and their rows, the filter's counts and
  stamps after every step; the eviction a checkpoint triggers; a restored twin continues bit-alike.
  data_seed: this model is chaotic in its first steps - fresh rows are ~0.0025 wide, Adam's first move is lr = 0.001
  whatever the gradient's size, and the first BatchNorm scales the lot to unit variance - so ONE ReLU input within
  rounding of zero flips an example's gradient, the fresh rows it touches move the other way and the next logits differ
  by 0.1 (the float32 and float64 oracles part the same way).  Of the seeds 12..17 the stand-in backend agrees with the
  oracle to 1e-6 over eight steps on 13, 15 and 17 and meets such a tie on the others.  value_steps: losses and row values
  are compared on the first that many steps (on the GPU, whose GEMMs round differently, the first three - a tie cannot
  have grown by then); the discrete state - admitted ids, counts, stamps, evictions - on every step."""
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.utils import checkpoint
  from oracle.model_oracle import OracleTrainer
  cfg = _filtered_config()
  est = EasyRecEstimator(cfg, device=device, batch_size=B, seed=4).build()
  tabs = est.engine.tables
  filt = {n: (tabs[n]['kv_filter_freq'], tabs[n]['kv_steps_to_live']) for n in est.engine.kv_tables}
  assert sorted(filt.values()) == [(0, 0), (0, 0), (0, 2), (0, 2), (2, 0), (2, 0), (3, 3), (3, 3)], filt  # (deep + wide)
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=B, seed=data_seed)
  batches = [gen.next_batch() for _ in range(steps + 3)]

  def step_both(i, b):
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert i >= value_steps or abs(got[k] - exp[k]) <= 1e-4 * max(1.0, abs(exp[k])), (i, k, got[k], exp[k])

  for i, b in enumerate(batches[:steps]):
    step_both(i, b)
    st = _compare_kv(est, orc, i, values=i < value_steps)
  # the counter filter keeps ids waiting: fewer rows than tracked ids on the filtered tables, none on the others
  for n, (ff, stl) in filt.items():
    if ff > 1:
      assert st[n + '/keys'].size < st[n + '/kv_seen_keys'].size, n
      waiting = ~np.isin(st[n + '/kv_seen_keys'], st[n + '/keys'])
      assert bool((st[n + '/kv_freq'][waiting] < ff).all()) and bool((st[n + '/kv_freq'][~waiting] == ff).all()), n
  # a checkpoint evicts what was not looked up for steps_to_live steps
  before = {n: st[n + '/kv_seen_keys'].size for n in filt if filt[n][1] > 0}
  ckpt = os.path.join(str(tmp_path), 'model.ckpt-%d' % est.global_step)
  checkpoint.save(est, ckpt)
  dropped = {n: orc.kv_evict(n) for n in est.engine.kv_tables}
  assert all(dropped[n] > 0 for n in before), dropped
  assert all(dropped[n] == 0 for n in filt if filt[n][1] == 0)
  st = _compare_kv(est, orc, 'evicted', values=steps <= value_steps)
  assert all(st[n + '/kv_seen_keys'].size == before[n] - dropped[n] for n in before)
  # a twin restored from the checkpoint and the (compacted) original continue alike, both like the oracle
  twin = EasyRecEstimator(cfg, device=device, batch_size=B, seed=99).build()
  checkpoint.restore(twin, ckpt)
  a, b = est.state_dict(slots=True), twin.state_dict(slots=True)
  for n in est.engine.kv_tables:
    for suffix in ('', '/keys', '/m', '/v') + (('/kv_seen_keys', '/kv_freq', '/kv_version') if (n + '/kv_freq') in a else ()):
      assert np.array_equal(a[n + suffix], b[n + suffix]), (n, suffix)
  for i, bt in enumerate(batches[steps:]):
    twin.train_step(bt)
    step_both(steps + i, bt)
    la, lb = est.loss_values(), twin.loss_values()
    assert all(abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(la[k])) for k in la), (i, la, lb)
    _compare_kv(est, orc, steps + i, values=steps + i < value_steps)
  return est


def test_kv_counter_filter_and_eviction_on_the_stand_in_backend(ref_backend, tmp_path):
  _run_filtered('cpu', tmp_path)


@pytest.mark.gpu
def test_kv_counter_filter_and_eviction_on_the_gpu(tmp_path):
  _run_filtered('cuda:0', tmp_path, value_steps=3)


@pytest.mark.gpu
def test_kv_filtered_translate_kernel():
  """The filtered insert against its python restatement (oracle/kernel_ref.py _kv_insert) over several launches with
  repeats inside and across launches: admitted ids, counts (clipped at filter_freq), stamps; export_all -> rebuild."""
  import torch

  from easyrec_amd import kernels
  from oracle.kernel_ref import RefBackend
  hip, ref, dev = kernels.hip(), RefBackend(), 'cuda:0'
  cap, dim, ff = 4096, 4, 3
  step_d, step_h = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64)
  var_d, var_h = torch.zeros(cap, dim, device=dev), torch.zeros(cap, dim)
  kd = hip.kv_create(var_d, cap, 77, 0.0, 0.5, filter_freq=ff, steps_to_live=4, step=step_d)
  kh = ref.kv_create(var_h, cap, 77, 0.0, 0.5, filter_freq=ff, steps_to_live=4, step=step_h)
  g = torch.Generator().manual_seed(5)
  for launch in range(6):
    step_d.fill_(launch + 1)
    step_h.fill_(launch + 1)
    ids = (torch.randint(0, 1500, (3000,), generator=g, dtype=torch.int64) ** 2) % 2000 * 7919 + 3
    ids[::11] = -1
    rd, rh = torch.empty(3000, dtype=torch.int64, device=dev), torch.empty(3000, dtype=torch.int64)
    hip.kv_translate(kd, ids.to(dev), rd, True)
    ref.kv_translate(kh, ids, rh, True)
    assert torch.equal(rd.cpu() >= 0, rh >= 0), launch
    a, b = hip.kv_export_all(kd), ref.kv_export_all(kh)
    assert torch.equal(a[0].cpu(), b[0]) and torch.equal(a[1].cpu() >= 0, b[1] >= 0)
    assert torch.equal(a[2].cpu().clamp(max=ff), b[2].clamp(max=ff)) and torch.equal(a[3].cpu(), b[3])
    assert int(kd['n_keys'].item()) == b[0].numel() and int(kd['next_row'].item()) == int((b[1] >= 0).sum())
  keys, rows, freq, version = hip.kv_export_all(kd)
  admitted = rows >= 0
  want = RefBackend.kv_init_value(77, keys[admitted].cpu().numpy(), dim, 0.0, 0.5)
  assert np.array_equal(var_d[rows[admitted]].cpu().numpy(), want)
  # rebuild from the exported records: the same lookups
  probe = keys[torch.randperm(keys.numel(), generator=g)[:500].to(dev)].contiguous()
  before = torch.empty(probe.numel(), dtype=torch.int64, device=dev)
  hip.kv_translate(kd, probe, before, False)
  hip.kv_rebuild(kd, keys, rows, freq, version)
  after = torch.empty(probe.numel(), dtype=torch.int64, device=dev)
  hip.kv_translate(kd, probe, after, False)
  assert torch.equal(before, after) and int(kd['overflow'].item()) == 0
  again = hip.kv_export_all(kd)
  assert all(torch.equal(x, y) for x, y in zip(again, (keys, rows, freq, version)))


@pytest.mark.gpu
def test_kv_translate_kernel():
  """er_kv_translate: every id gets ONE row however often and wherever it occurs in the launch, rows are the generator's
  values, ids < 0 and (without insert) unseen ids map to -1, the arena's end sets the overflow flag."""
  import torch

  from easyrec_amd import kernels
  from oracle.kernel_ref import RefBackend
  hip, dev = kernels.hip(), 'cuda:0'
  cap, dim = 5000, 8
  var = torch.zeros(cap, dim, device=dev)
  kv = hip.kv_create(var, cap, 1234, 0.1, 0.7)
  g = torch.Generator().manual_seed(3)
  ids = torch.randint(0, 3000, (20000,), generator=g, dtype=torch.int64) * 982451653 + 17
  ids[::7] = -1
  ids_d, rows = ids.to(dev), torch.empty(20000, dtype=torch.int64, device=dev)
  hip.kv_translate(kv, ids_d, rows, True)
  torch.cuda.synchronize()
  r = rows.cpu()
  assert bool((r[ids < 0] == -1).all()) and bool((r[ids >= 0] >= 0).all())
  uniq = torch.unique(ids[ids >= 0])
  assert int(kv['next_row'].item()) == uniq.numel() and int(kv['overflow'].item()) == 0
  by_id = {}
  for k, row in zip(ids.tolist(), r.tolist()):
    if k >= 0:
      assert by_id.setdefault(k, row) == row
  assert len(set(by_id.values())) == len(by_id)
  keys, krows = hip.kv_export(kv)
  assert torch.equal(keys.cpu(), uniq)
  want = RefBackend.kv_init_value(1234, keys.cpu().numpy(), dim, 0.1, 0.7)
  assert np.array_equal(var[krows].cpu().numpy(), want), 'rows are bit-identical to the numpy restatement of the generator'
  # without insert: known ids find their rows, unknown ones read -1
  probe = torch.cat([ids[:100], torch.tensor([5, 6, 7], dtype=torch.int64)]).to(dev)
  out = torch.empty(103, dtype=torch.int64, device=dev)
  hip.kv_translate(kv, probe, out, False)
  assert torch.equal(out[:100].cpu(), r[:100]) and bool((out[100:] == -1).all())
  # overflow
  more = (torch.arange(10000, dtype=torch.int64) + 10**12).to(dev)
  hip.kv_translate(kv, more, torch.empty(10000, dtype=torch.int64, device=dev), True)
  assert int(kv['overflow'].item()) == 1


@pytest.mark.gpu
def test_kv_embeddings_match_the_oracle_on_the_gpu():
  est, cfg, batches, st = _run('cuda:0')
  _evaluate_and_reload(est, cfg, batches, st, 'cuda:0')


@pytest.mark.gpu
def test_kv_tag_features_match_the_oracle_on_the_gpu():
  # (data seed: tools/scan_kv_mmoe_seeds_gpu.py - of the seeds 12..23, seven (13, 15-20) agree with the oracle on EVERY
  #  row over the three steps; the others put an example on a ReLU tie of an expert, whose pre-activations are not
  #  normalised - BatchNorm on the moving statistics, as the reference's MMoE - and a few rows then differ by O(lr))
  _run('cuda:0', config='mmoe_kv_taobao_small.config', n_kv=4, row_tol=2e-3, data_seed=16)


def _adagrad_kv_config(tmp_path):
  """deepfm_kv_criteo_small with an Adagrad embedding optimizer (initial_accumulator_value 0.2)"""
  text = open(os.path.join(ROOT, 'configs', 'deepfm_kv_criteo_small.config')).read()
  head = ('train_config {\n  optimizer_config {\n    adagrad_optimizer {\n      learning_rate {\n        '
          'constant_learning_rate {\n          learning_rate: 0.05\n        }\n      }\n      '
          'initial_accumulator_value: 0.2\n    }\n  }\n')
  assert text.count('train_config {\n') == 1
  path = os.path.join(str(tmp_path), 'deepfm_kv_adagrad.config')
  with open(path, 'w') as f:
    f.write(text.replace('train_config {\n', head))
  return config_util.get_configs_from_pipeline_file(path)


def _adagrad_restore_then_train(device, tmp_path):
  """Round-4 advisor finding: a restore (and an eviction) reset the unused arena rows' Adagrad accumulator to 0 instead of
  initial_accumulator_value, so ids created AFTER the restore took lr * sign(g) steps.  A restored twin must continue
  exactly like the uninterrupted run - on a batch full of ids neither has seen."""
  from easyrec_amd import kernels
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  cfg = _adagrad_kv_config(tmp_path)
  est = EasyRecEstimator(cfg, device=device, batch_size=64, seed=4).build()
  assert est.opt_emb.kind == kernels.OPT_ADAGRAD and abs(est.engine.slot_init['v'] - 0.2) < 1e-7
  gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=64, seed=21)
  for _ in range(2):
    est.train_step(gen.next_batch())
  twin = EasyRecEstimator(cfg, device=device, batch_size=64, seed=4).build()
  twin.load_state_dict(est.state_dict(slots=True))
  fresh = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=64, seed=77).next_batch()  # new ids
  est.train_step(fresh)
  twin.train_step(fresh)
  a, b = est.state_dict(slots=True), twin.state_dict(slots=True)
  n_new = 0
  for name in est.engine.kv_tables:
    assert np.array_equal(a[name + '/keys'], b[name + '/keys']), name
    assert np.array_equal(a[name + '/v'], b[name + '/v']), name   # the accumulators of old AND new ids
    assert np.array_equal(a[name], b[name]), name
    n_new += int((np.abs(a[name + '/v'] - 0.2).max(axis=1) > 0).sum())
    assert float(a[name + '/v'].min()) >= 0.2 - 1e-7, name        # no accumulator started from 0
  assert n_new > 50


def test_adagrad_kv_restore_then_train_on_the_stand_in_backend(ref_backend, tmp_path):
  _adagrad_restore_then_train('cpu', tmp_path)


@pytest.mark.gpu
def test_adagrad_kv_restore_then_train_on_the_gpu(tmp_path):
  _adagrad_restore_then_train('cuda:0', tmp_path)


def _din_with_hash_table_sequences(device, steps=3, B=48):
  """SequenceFeatures with `ev_params` (the last f4 leftover of round 4): the history sequences of MultiTowerDIN's attention
  (`tag_brand_list`, `tag_category_list`) embedded from hash-table tables - the sequence column creates its variable like any
  other (compat/feature_column/feature_column_v2.py:3616-3640 under :3478-3513), keyed by the hash into the whole int64
  range; padding positions own no row.  Losses of every step, the materialised ids and every row against the oracle."""
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'din_taobao_small.config'))
  seqs = [f for f in cfg.feature_config.features if f.feature_type == f.SequenceFeature]
  assert len(seqs) == 2
  for f in seqs:
    f.ev_params.max_capacity = 2048
  est = EasyRecEstimator(cfg, device=device, batch_size=B, seed=4).build()
  kv_names = sorted(est.engine.kv_tables)
  assert len(kv_names) == 2 and all('tag_' in n for n in kv_names), kv_names
  state0 = est.state_dict()
  orc = OracleTrainer(cfg, state0, batch_size=B)
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=12)
  for step in range(steps):
    b = gen.next_batch()
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      # (step 0 from identical parameters: the 1e-4 bar.  From the second Adam step on this 48-row BatchNorm model amplifies
      # any difference in fp32 summation order; the bars are tests/_bars.py's, each at most ten times what a one-ulp
      # perturbation of the oracle's own gradients produces at that step - measured by tests/test_chaos_bars.py on the CPU)
      assert abs(got[k] - exp[k]) <= _bars.bar('din_taobao_small', step) * max(1.0, abs(exp[k])), (step, k, got[k], exp[k])
    if step == min(steps, 2) - 1:
      # Adam's first moments row by row after the SECOND step: from the third on single rows' moments of this model differ by
      # tens of percent between two fp32 summation orders (the loss bar above is what the chaos envelope allows there)
      st2 = est.state_dict(slots=True)
      for n in kv_names:
        keys2, _ = orc.kv_state(n)
        assert np.array_equal(st2[n + '/keys'], keys2), n
        _, m_rows = orc.kv_state(n, orc.slots[n + '/m'])
        m_scale = float(np.abs(m_rows).max())
        assert float(np.abs(st2[n + '/m'] - m_rows).max()) <= 1e-3 * m_scale + 1e-9, n
  st = est.state_dict(slots=True)
  for n in kv_names:
    keys, rows = orc.kv_state(n)
    assert np.array_equal(st[n + '/keys'], keys), n
    assert keys.size > 20 and st[n].shape == rows.shape
    assert float(np.abs(st[n] - rows).max()) <= 4e-3 * steps, n   # (a few lr-sized steps: Adam on near-zero gradients)
  # evaluation creates no rows, and a restored twin continues like the original
  n_before = {n: st[n + '/keys'].size for n in kv_names}
  est.evaluate([SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=99).next_batch()])
  after = est.state_dict()
  assert all(after[n + '/keys'].size == n_before[n] for n in kv_names)
  twin = EasyRecEstimator(cfg, device=device, batch_size=B, seed=4).build()
  twin.load_state_dict(est.state_dict(slots=True))
  twin.set_global_step(est.global_step)
  b = gen.next_batch()
  est.train_step(b)
  twin.train_step(b)
  a, c = est.loss_values(), twin.loss_values()
  assert all(abs(a[k] - c[k]) <= 1e-6 * max(1.0, abs(a[k])) for k in a), (a, c)


def test_hash_table_sequence_features_match_the_oracle_on_the_stand_in_backend(ref_backend):
  _din_with_hash_table_sequences('cpu')


@pytest.mark.gpu
def test_hash_table_sequence_features_match_the_oracle_on_the_gpu():
  # three steps under tests/_bars.py's per-step bars (1e-4, 1e-4, 1e-2): the third step's bar is what test_chaos_bars.py
  # justifies - a one-ulp change of the oracle's own gradients moves that step's loss by up to 2.8e-3 over six model seeds
  _din_with_hash_table_sequences('cuda:0', steps=3)
