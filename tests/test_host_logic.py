"""Host logic on CPU: the package's models/estimator driven through the per-kernel CPU oracle
(monkeypatched backend) must reproduce the independent model-level oracle.  This checks the HOST
code (plan building, variable naming, loss assembly, optimizer scalars) - the kernels are checked by
the -m gpu tests."""
import logging
import os

import numpy as np
import pytest

from easyrec_amd.builders import optimizer_builder
from easyrec_amd.utils import config_util

logging.disable(logging.WARNING)
BN_BIAS = '/bias'  # biases followed by BatchNorm have an identically-zero gradient


def _compare_states(est_state, oracle_state, skip_bn_bias=True, tol=3e-4):
  worst = ('', 0.0)
  for k, v in oracle_state.items():
    if k not in est_state:
      continue
    if skip_bn_bias and k.endswith(BN_BIAS) and (k[:-len('/bias')] + '/bn/gamma') in oracle_state:
      continue
    a, b = est_state[k], v
    if k.endswith('/bn/moving_mean'):
      # the moving mean tracks mean(x @ W + bias): compare it net of the (noise-driven) bias
      bias = k[:-len('/bn/moving_mean')] + '/bias'
      a, b = a - 0.01 * est_state[bias], b - 0.01 * oracle_state[bias]
      continue
    d = float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12))
    if d > worst[1]:
      worst = (k, d)
  assert worst[1] < tol, worst


@pytest.mark.parametrize('config', ['deepfm_criteo_small.config'])
def test_estimator_matches_model_oracle(ref_backend, config):
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file(os.path.join('configs', config))
  B = 48
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  gen = SyntheticCriteo(cfg.data_config, est.feature_configs, batch_size=B)
  for _ in range(3):
    b = gen.next_batch()
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(got[k] - exp[k]) <= 1e-5 * max(1.0, abs(exp[k])), (k, got[k], exp[k])
  est.varstore.check_grad_views()
  _compare_states(est.state_dict(), orc.state)


def test_variable_names_follow_tf_scopes(ref_backend):
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo_small.config')
  est = EasyRecEstimator(cfg, device='cpu', batch_size=8).build()
  names = set(est.state_dict())
  for n in ('input_layer/C1_embedding/embedding_weights',  # wide group is called first
            'input_layer_1/C1_embedding/embedding_weights',
            'input_layer_1/F1_weighted_by_F1_raw_proj_val_embedding/embedding_weights',
            'deep_feature/dnn_0/kernel', 'deep_feature/dnn_0/bn/moving_variance', 'final_dnn/dnn_2/bias',
            'output/kernel'):
    assert n in names, n
  assert est.engine.tables['input_layer/C1_embedding/embedding_weights']['dim'] == 1
  assert est.engine.tables['input_layer_1/C1_embedding/embedding_weights']['dim'] == 16
  # dense_regularization (deprecated alias) is the kernels' L2, biases carry none
  assert est.varstore.l2_of('deep_feature/dnn_0/kernel') == pytest.approx(1e-5)
  assert est.varstore.l2_of('deep_feature/dnn_0/bias') == 0.0


def test_adam_scalars_follow_tf():
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo.config')
  st = optimizer_builder.build(cfg.train_config.optimizer_config[0])
  row = st.hyper_row(0)
  f = np.float32
  assert row[0] == f(0.001)
  assert row[1] == f(0.001) * np.sqrt(f(1) - f(0.999)) / (f(1) - f(0.9))
  for _ in range(1000):
    st.finish_step()
  assert st.hyper_row(999)[0] == f(0.001) and st.hyper_row(1000)[0] == f(0.0005)
  assert st.hyper_row(10**6)[0] == f(1e-5)  # min_learning_rate


def test_csv_input_roundtrip(tmp_path, built_lib):
  from easyrec_amd.input.csv_input import CSVInput
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo_small.config')
  rows = []
  rng = np.random.default_rng(0)
  for i in range(4):
    f = ['%d' % rng.integers(0, 50) if rng.random() > 0.2 else '' for _ in range(13)]
    c = ['%08x' % rng.integers(0, 2**32) if rng.random() > 0.2 else '' for _ in range(26)]
    rows.append('\t'.join(['%d' % (i % 2)] + f + c))
  p = tmp_path / 'data.tsv'
  p.write_text('\n'.join(rows) + '\n')
  feats = list(cfg.feature_config.features)
  inp = CSVInput(cfg.data_config, feats, str(p), batch_size=4, hash_on_host=True)
  batch = next(inp.batches())
  assert batch['labels'].tolist() == [[0.0, 1.0, 0.0, 1.0]]
  assert batch['raw'].shape == (13, 4) and batch['hash_ids'].shape == (26, 4)
  # '' -> dropped (-1); non-empty -> FarmHash bucket
  from oracle import hashing
  first = rows[0].split('\t')
  for j in range(26):
    s = first[14 + j]
    exp = -1 if s == '' else hashing.fingerprint64(s) % 1000
    assert batch['hash_ids'][j, 0] == exp
  # F2: (x - (-3)) / (257675 + 3) in fp32; '' -> default 0
  x = np.float32(float(first[2]) if first[2] else 0.0)
  assert batch['raw'][1, 0] == (x - np.float32(-3.0)) / np.float32(257678.0)


@pytest.mark.parametrize('config,B', [('dcn_criteo_small.config', 32), ('din_taobao_small.config', 24),
                                      ('mmoe_taobao_small.config', 24), ('dcn_v2_criteo_small.config', 32),
                                      ('dcn_v2_lowrank_criteo_small.config', 32),
                                      ('wide_and_deep_criteo_small.config', 24),
                                      ('wide_and_deep_nofinal_criteo_small.config', 24), ('fm_criteo_small.config', 24),
                                      ('multi_tower_criteo_small.config', 24), ('dlrm_criteo_small.config', 24),
                                      ('dlrm_itself_criteo_small.config', 24), ('dlrm_cat_criteo_small.config', 24),
                                      ('deepfm_bucketized_criteo_small.config', 24),
                                      ('dlrm_shared_criteo_small.config', 24), ('deepfm_shared_criteo_small.config', 24),
                                      ('deepfm_combo_criteo_small.config', 24), ('deepfm_lookup_criteo_small.config', 24),
                                      ('simple_multi_task_taobao_small.config', 24), ('ple_taobao_small.config', 24),
                                      ('dbmtl_taobao_small.config', 24), ('dbmtl_mmoe_taobao_small.config', 24),
                                      ('mmoe_backbone_taobao_small.config', 24),
                                      ('multi_tower_f1_pairwise_criteo_small.config', 24),
                                      ('deepfm_adagrad_criteo_small.config', 24),
                                      ('mmoe_tower_losses_taobao_small.config', 24),
                                      ('dbmtl_numeric_sequences_taobao_small.config', 24),
                                      ('dbmtl_numeric_sequences_dnn_taobao_small.config', 24),
                                      # (B = 24 with this data seed puts one example of cvr/dnn_0 on a ReLU tie: B = 32)
                                      ('mmoe_backbone_bayes_taobao_small.config', 32)])
def test_other_models_match_model_oracle(ref_backend, config, B):
  """DCN / MultiTowerDIN / MMoE host logic (variable naming, layer wiring, multi-task losses, sequence and
  tag lookups) against the independent model-level oracle, 2 optimisation steps."""
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file(os.path.join('configs', config))
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=11)
  for step in range(2):
    b = gen.next_batch()
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(got[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, got[k], exp[k])
    if step == 0:
      # gradients of every variable, read back as Adam's first moment m = (1 - beta1) * g (the parameters
      # themselves go through Adam's normalisation, which amplifies rounding noise of near-zero gradients)
      st = est.state_dict(slots=True)
      names = set(orc.state)
      n_cmp = 0
      # tensors whose true gradient is zero (e.g. the last cross bias in front of a BatchNorm) carry only
      # rounding noise: allow 1e-6 of the largest gradient scale in the model
      gmax = max(float(np.max(np.abs(v))) for kk, v in orc.slots.items() if kk.endswith('/m'))
      for k in orc.state:
        key = k + '/m'
        if key not in orc.slots or key not in st:
          continue
        if k.endswith('/bias') and (k[:-len('/bias')] + '/bn/gamma') in names:
          continue  # d(loss)/d(bias) == 0 under BatchNorm
        ref = orc.slots[key]
        d, scale = float(np.max(np.abs(st[key] - ref))), float(np.max(np.abs(ref)))
        assert d <= 2e-4 * scale + 1e-6 * gmax, (key, d, scale)
        n_cmp += 1
      assert n_cmp > 5
  est.varstore.check_grad_views()


@pytest.mark.parametrize('config', ['deepfm_criteo_small.config', 'mmoe_taobao_small.config'])
def test_packed_batch_loads_like_the_plain_one(ref_backend, config):
  """DeviceFeatures.pack(): the fixed-size inputs as ONE byte image of the input arena (a single copy per step);
  loading it must leave every buffer exactly as the per-array load does, ragged parts included."""
  import torch
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  cfg = config_util.get_configs_from_pipeline_file(os.path.join('configs', config))
  B = 24
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=1).build()
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=5)
  f = est.features
  for _ in range(2):
    batch = gen.next_batch()
    f.load(batch)
    plain = f.arena.clone()
    tags = {k: {n: (None if t is None else t.clone()) for n, t in v.items()} for k, v in f.tags.items()}
    f.arena.zero_()
    packed = f.pack(batch)
    assert set(packed) & {'labels', 'raw', 'str_bytes', 'str_offsets', 'hash_ids', 'int_ids'} == set()
    f.load(packed)
    for name, (o, nbytes, _, _) in f._layout.items():
      if name == 'hash_ids' and packed['packed_has_strings']:
        continue  # produced on the device by transform()
      assert torch.equal(f.arena[o:o + nbytes], plain[o:o + nbytes]), name
    for k, v in f.tags.items():
      for n, t in v.items():
        assert t is None or torch.equal(t, tags[k][n]), (k, n)


def test_bucketized_raw_feature_ids():
  """RawFeature with boundaries / num_buckets -> BucketizedColumn ids (reference feature_column/feature_column.py:
  365-386, compat/feature_column/feature_column_v2.py:2762-2916): bucket = number of boundaries <= the NORMALISED
  value; a value equal to a boundary belongs to the upper bucket."""
  from easyrec_amd.input.csv_input import CSVInput
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_bucketized_criteo_small.config')
  B = 6
  reader = CSVInput(cfg.data_config, cfg.feature_config.features, None, batch_size=B)
  sch = reader.schema
  names = [x.input_name for x in cfg.data_config.input_fields]
  cols = {n: (['0'] * B if not n.startswith('C') else ['ab'] * B) for n in names}
  f1 = {f.input_names[0]: f for f in cfg.feature_config.features}['F1']
  span = f1.max_val - f1.min_val
  # normalised 0.0, 0.05 (on a boundary), 0.1, 0.5 (boundary), 0.79, 1.0 -> buckets 0, 1, 1, 3, 3, 4
  cols['F1'] = [str(f1.min_val + v * span) for v in (0.0, 0.05, 0.1, 0.5, 0.79, 1.0)]
  batch = reader.preprocess(cols)
  got = batch['int_ids'][sch.int_single['F1']['col']]
  x = batch['raw'][sch.raw['F1']['row']]
  exp = [sum(1 for b in (0.05, 0.2, 0.5, 0.8) if np.float32(b) <= v) for v in x]
  assert list(got) == exp and exp[0] == 0 and exp[3] == 3 and exp[5] == 4, (list(got), exp, list(x))
  assert sch.int_single['F3']['num_buckets'] == 11 and sch.int_single['F1']['num_buckets'] == 5


def test_combo_feature_crossed_column_through_the_input_and_the_model(ref_backend, tmp_path):
  """ComboFeature without combo_join_sep = crossed_column (reference feature_column.py:434-445): CSVInput computes the
  crossed id (sparse_cross_hashed of the two inputs' strings, '' included) and the model looks it up like any id
  column; ids against the pinned restatement, two training steps against the model oracle."""
  from easyrec_amd.input.csv_input import CSVInput
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle import hashing
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_combo_criteo_small.config')
  B, rows = 24, []
  rng = np.random.default_rng(2)
  for i in range(2 * B):
    f = ['%d' % rng.integers(0, 50) if rng.random() > 0.2 else '' for _ in range(13)]
    c = ['%02x' % rng.integers(0, 40) if rng.random() > 0.15 else '' for _ in range(26)]
    rows.append('\t'.join(['%d' % (i % 3 == 0)] + f + c))
  p = tmp_path / 'data.tsv'
  p.write_text('\n'.join(rows) + '\n')
  feats = list(cfg.feature_config.features)
  inp = CSVInput(cfg.data_config, feats, str(p), batch_size=B, hash_on_host=True)
  batches = list(inp.batches())[:2]
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  col = est.features.schema.int_single['C1_C2_cross']['col']
  got = batches[0]['int_ids'][col]
  for r in range(B):
    c1, c2 = rows[r].split('\t')[14], rows[r].split('\t')[15]
    assert got[r] == hashing.sparse_cross_hashed([c1, c2], 1000), (r, c1, c2)  # ('' is crossed like any value)
  assert (got >= 0).all()
  assert any('C1_C2_cross' in n for n in est.state_dict())  # its own embedding tables (deep and wide)
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  for b in batches:
    est.train_step(b)
    res, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(res[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, res[k], exp[k])
  st = est.state_dict()
  for k, v in orc.state.items():
    if 'C1_C2_cross' in k:
      assert np.allclose(st[k], v, rtol=1e-4, atol=1e-6), k


def test_joined_combo_feature_through_the_input_and_the_model(ref_backend, tmp_path):
  """ComboFeature WITH combo_join_sep = one hashed column over the inputs' strings joined by the separator (reference
  input/input.py:425-430 string_join, feature_column/feature_column.py:446-455; the shape of
  samples/model_config/deepfm_combo_v2_on_avazu_ctr.config): CSVInput joins and hashes, ids against the pinned hash
  restatement, two training steps against the model oracle."""
  from easyrec_amd.input.csv_input import CSVInput
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle import hashing
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_combo_criteo_small.config')
  combo = [f for f in cfg.feature_config.features if f.feature_name == 'C1_C2_cross'][0]
  combo.combo_join_sep = 'X'
  B, rows = 24, []
  rng = np.random.default_rng(5)
  for i in range(2 * B):
    f = ['%d' % rng.integers(0, 50) if rng.random() > 0.2 else '' for _ in range(13)]
    c = ['%02x' % rng.integers(0, 40) if rng.random() > 0.15 else '' for _ in range(26)]
    rows.append('\t'.join(['%d' % (i % 3 == 0)] + f + c))
  p = tmp_path / 'data.tsv'
  p.write_text('\n'.join(rows) + '\n')
  inp = CSVInput(cfg.data_config, list(cfg.feature_config.features), str(p), batch_size=B, hash_on_host=True)
  batches = list(inp.batches())[:2]
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  assert 'C1_C2_cross' not in est.features.schema.int_single
  col = est.features.schema.hash_single['C1_C2_cross']['col']
  got = batches[0]['hash_ids'][col]
  for r in range(B):
    joined = (rows[r].split('\t')[14] + 'X' + rows[r].split('\t')[15]).encode()
    want = hashing.hash_bucket_fast(np.frombuffer(joined, dtype=np.uint8), np.array([0, len(joined)]), 1, [1000], True)
    assert got[r] == want[0], (r, joined)  # (both inputs empty is still the string 'X': never dropped)
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  for b in batches:
    est.train_step(b)
    res, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(res[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, res[k], exp[k])
  st = est.state_dict()
  for k, v in orc.state.items():
    if 'C1_C2_cross' in k:
      assert np.allclose(st[k], v, rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize('loss_type', ['L2_LOSS', 'SIGMOID_L2_LOSS'])
def test_regression_heads_match_model_oracle(ref_backend, loss_type):
  """loss_type L2_LOSS / SIGMOID_L2_LOSS (reference model/rank_model.py:123-128: y = the output column, through a sigmoid
  for the latter; builders/loss_builder.py:52-55 mean_squared_error; the shape of
  samples/model_config/deepfm_combo_on_avazu_reg.config): losses, predictions' key and the updated variables of two
  steps against the model oracle."""
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.protos.loss_pb2 import LossType
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo_small.config')
  cfg.model_config.loss_type = LossType.Value(loss_type)
  B = 24
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=11)
  for _ in range(2):
    b = gen.next_batch()
    est.train_step(b)
    got, exp = est.loss_values(), orc.train_step(b)
    assert sorted(got) == sorted(exp) and 'l2_loss' in exp
    for k in exp:
      assert abs(got[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, got[k], exp[k])
  st = est.state_dict()
  for k, v in orc.state.items():
    if not k.endswith(('kernel', 'embedding_weights', 'gamma', 'beta')):
      continue  # (a bias under BatchNorm has d(loss)/d(bias) == 0: Adam normalises pure rounding noise, and the moving
      # mean follows that bias)
    d = float(np.max(np.abs(st[k] - v)))
    assert d <= 1e-4, (k, d)  # (a tenth of one Adam step: Adam's normalisation amplifies the rounding of small gradients)


def test_lookup_feature_through_the_input_and_the_model(ref_backend, tmp_path):
  """LookupFeature (reference input/input.py:941-1000): the values of the row's map whose key equals the row's key,
  hashed, combined ('mean' here) - ids against a direct restatement, two training steps against the model oracle."""
  from easyrec_amd.input.csv_input import CSVInput
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle import hashing
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_lookup_criteo_small.config')
  B, rows, want = 24, [], []
  rng = np.random.default_rng(4)
  for i in range(2 * B):
    f = ['%d' % rng.integers(0, 50) if rng.random() > 0.2 else '' for _ in range(13)]
    c = ['k%d' % rng.integers(0, 4) for _ in range(26)]
    pairs = [('k%d' % rng.integers(0, 4), 'v%d' % rng.integers(0, 30)) for _ in range(int(rng.integers(0, 7)))]
    while sum(1 for k, _ in pairs if k == c[2]) > 4:  # lookup_max_sel_elem_num
      pairs.pop()
    kv = '|'.join('%s:%s' % p for p in pairs)
    want.append([v for k, v in pairs if k == c[2]])
    rows.append('\t'.join(['%d' % (i % 3 == 0)] + f + c + [kv]))
  p = tmp_path / 'data.tsv'
  p.write_text('\n'.join(rows) + '\n')
  inp = CSVInput(cfg.data_config, list(cfg.feature_config.features), str(p), batch_size=B, hash_on_host=True)
  batches = list(inp.batches())[:2]
  ids, offs = batches[0]['tag/C3_lookup/ids'], batches[0]['tag/C3_lookup/offsets']
  assert offs[-1] == len(ids) and len(ids) > B // 4
  for r in range(B):
    got = ids[offs[r]:offs[r + 1]].tolist()
    assert got == [hashing.fingerprint64(v) % 500 for v in want[r]], r
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  for b in batches:
    est.train_step(b)
    res, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(res[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, res[k], exp[k])
  st = est.state_dict()
  for k, v in orc.state.items():
    if 'C3_lookup' in k:
      assert np.allclose(st[k], v, rtol=1e-4, atol=1e-6), k


def test_combo_feature_with_multi_valued_inputs(ref_backend, tmp_path):
  """ComboFeature + combo_input_seps (input.py:383-407): the second input is split by '|'; every (C1 value, C4 token)
  pair is one crossed id - a ragged lookup.  Ids against the pinned restatement, two steps against the model oracle."""
  import itertools
  from easyrec_amd.input.csv_input import CSVInput
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from easyrec_amd.protos.feature_config_pb2 import FeatureConfig
  from oracle import hashing
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file('configs/deepfm_criteo_small.config')
  fc = cfg.feature_config.features.add()
  fc.input_names.extend(['C1', 'C4'])
  fc.combo_input_seps.extend(['', '|'])
  fc.feature_name, fc.feature_type, fc.hash_bucket_size, fc.embedding_dim = 'C1_C4_cross', FeatureConfig.ComboFeature, 700, 16
  fc.combiner = 'mean'
  for g in cfg.model_config.feature_groups:
    g.feature_names.append('C1_C4_cross')
  B, rows, want = 24, [], []
  rng = np.random.default_rng(6)
  for i in range(2 * B):
    f = ['%d' % rng.integers(0, 50) if rng.random() > 0.2 else '' for _ in range(13)]
    c = ['%02x' % rng.integers(0, 40) if rng.random() > 0.15 else '' for _ in range(26)]
    toks = ['t%d' % rng.integers(0, 9) for _ in range(int(rng.integers(0, 4)))]
    c[3] = '|'.join(toks)
    want.append(list(itertools.product([c[0]], toks)))
    rows.append('\t'.join(['%d' % (i % 3 == 0)] + f + c))
  p = tmp_path / 'data.tsv'
  p.write_text('\n'.join(rows) + '\n')
  inp = CSVInput(cfg.data_config, list(cfg.feature_config.features), str(p), batch_size=B, hash_on_host=True)
  batches = list(inp.batches())[:2]
  ids, offs = batches[0]['tag/C1_C4_cross/ids'], batches[0]['tag/C1_C4_cross/offsets']
  for r in range(B):
    assert ids[offs[r]:offs[r + 1]].tolist() == [hashing.sparse_cross_hashed(list(cb), 700) for cb in want[r]], r
  assert any(len(w) == 0 for w in want[:B]) and any(len(w) > 1 for w in want[:B])
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  for b in batches:
    est.train_step(b)
    res, exp = est.loss_values(), orc.train_step(b)
    for k in exp:
      assert abs(res[k] - exp[k]) <= 2e-5 * max(1.0, abs(exp[k])), (k, res[k], exp[k])


def test_bench_helpers_describe_every_baseline_config(ref_backend):
  """bench.py's host-side helpers (no GPU): which generator a config gets, the workload text of the JSON line, the ring
  source, and that a packed generic batch round-trips through DeviceFeatures on the stand-in backend."""
  import importlib.util
  import types
  ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  criteo = {'deepfm_criteo.config': True, 'dcn_v2_criteo.config': True, 'din_taobao_10m.config': False,
            'mmoe_taobao_4task_d64_25m.config': False, 'xdeepfm_taobao.config': False}
  for name, want in criteo.items():
    cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', name))
    assert bench.is_criteo_shaped(cfg) == want, name
    args = types.SimpleNamespace(config=os.path.join(ROOT, 'configs', name), dense_dtype='f32', ids='zipf', precondition=8)
    est = types.SimpleNamespace(opt_emb=types.SimpleNamespace(name='adam_optimizer'), dense_sweep=False)
    text = bench.workload_text(cfg, args, est, want, cfg.data_config.batch_size, 'hipGraph replay', 4)
    assert name in text and 'batch %d' % cfg.data_config.batch_size in text
    assert ('Criteo' in text) == want and ('Taobao' in text) == (not want)
  ring = bench.RingSource(['a', 'b', 'c'])
  assert [ring.next_packed() for _ in range(4)] == ['b', 'c', 'a', 'b']
  assert 'examples/sec' in bench.baseline_metric()


def test_fm_over_fields_of_unequal_widths_is_rejected(ref_backend):
  """layers/fm.py:20 stacks the fields (tf.stack): fields of different embedding_dim fail when the reference builds the
  graph (the shape of examples/configs/fm_on_criteo.config: embedding_dim 10 beside 16).  The product raises too instead
  of reading the concatenation as F fields of the first width."""
  import torch

  from easyrec_amd.layers.fm import FM
  with pytest.raises(ValueError, match='same embedding_dim'):
    FM('fm')([torch.zeros(4, 10), torch.zeros(4, 16), torch.zeros(4, 10)])
