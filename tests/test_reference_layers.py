"""The oracle's restatements (and, on the GPU, the HIP kernels) against tests/golden/reference_layer_vectors.npz:
outputs of the REFERENCE'S OWN layer code - layers/fm.py FM, keras FM / DotInteraction / Cross / CIN
(layers/keras/interaction.py), core/learning_schedules.py exponential_decay_with_burnin - executed unmodified on a numpy
stand-in for the tensorflow module (tests/golden/make_reference_layer_vectors.py, run where /root/reference exists).
What this pins: index conventions of the reference code (summed axes, CIN's kernel[n, h, m] against x_k[h] and x_0[m],
the lower-triangle order of the dot interaction, diag_scale placement, staircase flooring) - not TensorFlow's rounding."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'reference_layer_vectors.npz'))


def _t(name, dtype=torch.float64):
  return torch.from_numpy(np.asarray(G[name])).to(dtype)


def _close(got, want, tol=1e-9):
  got, want = torch.as_tensor(got).double().cpu(), torch.as_tensor(want).double()
  assert got.shape == want.shape, (tuple(got.shape), tuple(want.shape))
  assert float((got.detach() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))


def test_fm_restatements():
  from oracle.kernel_ref import RefBackend
  x = _t('fm_inputs')  # [B, F, D]
  B, F, D = x.shape
  fm, _ = RefBackend().fm_fwd(x.reshape(B, F * D), F, D)
  _close(fm, _t('fm_layers_fm'))
  _close(fm, _t('fm_keras_variant'))
  _close(fm.sum(dim=1, keepdim=True), _t('fm_keras'))


def _keras_dot_index(F, self_interaction):
  """position of pair (i, j), i >= j, in the keras DotInteraction output (lower triangle, row-major: boolean_mask)"""
  order = [(i, j) for i in range(F) for j in range(i + 1 if self_interaction else i)]
  return {p: k for k, p in enumerate(order)}


@pytest.mark.parametrize('self_interaction', [False, True])
def test_dot_interaction_restatement(self_interaction):
  """The DLRM model class lists the pairs of the UPPER triangle row by row (model/dlrm.py:51-57, what this package's
  DLRM and er_dot_interaction implement); the keras DotInteraction layer (not exported here) lists the lower triangle.
  Same products: pair (i, j) of the one is pair (j, i) of the other."""
  from oracle.kernel_ref import RefBackend
  x = _t('fm_inputs')
  B, F, D = x.shape
  ref = RefBackend()
  got = ref.dot_interaction_fwd(x.reshape(B, F * D), F, D, self_interaction)
  keras = _t('dot_self%d_skip0' % int(self_interaction))
  where = _keras_dot_index(F, self_interaction)
  pairs = ref._dot_pairs(F, self_interaction)
  assert len(pairs) == keras.shape[1]
  _close(got, torch.stack([keras[:, where[(j, i)]] for i, j in pairs], dim=1))
  # skip_gather: the full F x F matrix with the other triangle zeroed - the same numbers in place
  full = _t('dot_self%d_skip1' % int(self_interaction)).reshape(B, F, F)
  keep = torch.tril(torch.ones(F, F), 0 if self_interaction else -1).bool()
  _close(full[:, keep], keras)


def test_keras_dot_interaction_layer_on_the_stand_in_backend(ref_backend):
  """the product's keras DotInteraction block (a column permutation of er_dot_interaction's output) in the reference
  layer's own order"""
  from easyrec_amd.layers.keras import DotInteraction

  class P(object):
    def __init__(self, **kw):
      self.kw = kw

    def get_or_default(self, k, d):
      return self.kw.get(k, d)

  x = _t('fm_inputs', torch.float32)
  feats = [x[:, i, :].contiguous() for i in range(x.shape[1])]
  for si in (False, True):
    got = DotInteraction(P(self_interaction=si), name='dot')(feats)
    _close(got, _t('dot_self%d_skip0' % int(si)), 1e-5)


def _vars(state):
  from oracle.model_oracle import Vars
  return Vars({k: np.asarray(v, dtype=np.float64) for k, v in state.items()}, torch.float64)


@pytest.mark.parametrize('tag', ['full', 'diag', 'lowrank'])
def test_cross_v2_restatement(tag):
  from google.protobuf import struct_pb2

  from oracle.model_oracle import OracleTrainer
  st = struct_pb2.Struct()
  state = {}
  if tag == 'lowrank':
    st['projection_dim'] = 3
    state['c/dense_u/kernel'], state['c/dense_v/kernel'] = G['cross_lowrank_u'], G['cross_lowrank_v']
    state['c/dense/bias'] = G['cross_lowrank_bias']
  else:
    state['c/dense/kernel'], state['c/dense/bias'] = G['cross_%s_kernel' % tag], G['cross_%s_bias' % tag]
    if tag == 'diag':
      st['diag_scale'] = 0.25
  got = OracleTrainer._keras_cross(None, _vars(state), _t('cross_x0'), _t('cross_x'), st, 'c')
  _close(got, _t('cross_%s_out' % tag))


def test_cross_v2_kernel_ref_without_bias():
  from oracle.kernel_ref import RefBackend
  x0, x = _t('cross_x0'), _t('cross_x')
  _close(RefBackend().cross_v2_fwd(x0, x, x @ _t('cross_nobias_kernel'), None, 0.0), _t('cross_nobias_out'))
  _close(RefBackend().cross_v2_fwd(x0, x, x @ _t('cross_diag_kernel'), _t('cross_diag_bias'), 0.25), _t('cross_diag_out'))


def test_cin_restatement():
  from oracle.model_oracle import OracleTrainer
  state = {'cin/cin_kernel_%d' % i: G['cin_kernel_%d' % i] for i in range(2)}
  state.update({'cin/cin_bias_%d' % i: G['cin_bias_%d' % i] for i in range(2)})
  got = OracleTrainer._keras_cin(None, _vars(state), _t('cin_x'), [5, 2], 'cin')
  _close(got, _t('cin_out'))


@pytest.mark.parametrize('tag', ['plain', 'burnin', 'smooth'])
def test_learning_rate_schedule(tag):
  from easyrec_amd.builders.optimizer_builder import exponential_decay_with_burnin
  base, decay_steps, factor, burn_lr, burn_steps, min_lr, staircase = [float(v) for v in G['lr_%s_args' % tag]]
  for step, want in zip(G['lr_steps'], G['lr_%s' % tag]):
    got = exponential_decay_with_burnin(int(step), base, int(decay_steps), factor, burnin_learning_rate=burn_lr,
                                        burnin_steps=int(burn_steps), min_learning_rate=min_lr, staircase=bool(staircase))
    assert abs(float(got) - float(want)) <= 2e-6 * float(want) + 1e-12, (tag, int(step), float(got), float(want))


def _layer_vars():
  """the variables the reference's tf.layers.* calls created, + the moving statistics the oracle's BatchNorm reads"""
  state = {k[len('var:'):]: G[k] for k in G.files if k.startswith('var:')}
  for k in list(state):
    if k.endswith('/bn/gamma'):
      n = state[k].shape[0]
      state[k[:-len('gamma')] + 'moving_mean'] = np.zeros(n)
      state[k[:-len('gamma')] + 'moving_variance'] = np.ones(n)
    if k.startswith('alpha_'):  # Dice: the statistics-only BatchNorm's moving averages
      n, base = state[k].shape[0], k[len('alpha_'):] + '/batch_normalization/'
      state[base + 'moving_mean'], state[base + 'moving_variance'] = np.zeros(n), np.ones(n)
  return _vars(state)


def _oracle():
  from oracle.model_oracle import OracleTrainer
  orc = OracleTrainer.__new__(OracleTrainer)
  orc._moving = {}
  return orc


class _DnnCfg(object):  # the fields of dnn_pb2.DNN the oracle reads

  def __init__(self, hidden_units, activation='tf.nn.relu'):
    self.hidden_units, self.use_bn, self.activation, self.dropout_ratio = hidden_units, True, activation, []


def test_dnn_din_mmoe_restatements():
  """layers/dnn.py DNN (dense -> BatchNorm over every axis but the last -> relu; the last layer optionally plain),
  model/multi_tower_din.py din() (concat order [q, h, q-h, q*h], mask value, softmax over time, output [pooled, q]) and
  layers/mmoe.py MMOE (experts stacked on axis 1, softmax gate over experts) - the reference's code on the stand-in."""
  orc, V = _oracle(), _layer_vars()
  _close(orc.dnn(V, _t('dnn_x'), _DnnCfg([6, 3]), 'tower', 0.0), _t('dnn_out'))
  _close(orc.dnn(V, _t('dnn_x'), _DnnCfg([6, 3]), 'tower2', 0.0, last_no_act=True, last_no_bn=True), _t('dnn_out_last_plain'))
  fea = {'key': _t('din_key'), 'hist_seq_emb': _t('din_hist'), 'hist_seq_len': torch.from_numpy(np.asarray(G['din_len']))}
  _close(orc._din(V, _DnnCfg([8, 4, 1]), fea, 'din', 0.0), _t('din_out'))
  tasks = orc._mmoe_layer(V, _t('mmoe_x'), [_DnnCfg([5, 3], 'relu')] * 3, 2, 0.0, training=True)  # (as generated)
  _close(tasks[0], _t('mmoe_task_0'))
  _close(tasks[1], _t('mmoe_task_1'))
  _close(orc._cross_net(V, _t('dcn_x'), 3), _t('dcn_cross_out'))  # model/dcn.py _cross_net


@pytest.mark.parametrize('tag,text', [('default', 'hidden_units: [6, 3]'),
                                      ('final_linear', "hidden_units: [4, 1] use_final_bn: false final_activation: 'linear'"),
                                      ('biased', 'hidden_units: [4, 2] use_bias: true use_final_bias: true use_bn: false'),
                                      ('final_none', "hidden_units: [5, 2] final_activation: ''")])
def test_keras_mlp_restatement(tag, text):
  """layers/keras/blocks.py MLP configured as a backbone block is (the reference's Parameter over dnn_pb2.MLP): Dense
  without bias -> BatchNorm -> activation per layer by default, the LAST layer with BatchNorm too and - the proto
  default of the string field coming through Parameter.get_or_default - ReLU, unless final_activation says otherwise
  ('' = none); use_final_bn / use_bias / use_final_bias / use_bn as configured."""
  from google.protobuf import text_format

  from easyrec_amd.protos import dnn_pb2
  cfg = dnn_pb2.MLP()
  text_format.Merge(text, cfg)
  _close(_oracle()._keras_mlp(_layer_vars(), _t('mlp_x'), cfg, 'mlp_%s' % tag, 0.0), _t('mlp_%s_out' % tag))


@pytest.mark.parametrize('tag,normalizer,need_target,query', [('softmax', 'softmax', True, 'kdin_query'),
                                                              ('sigmoid', 'sigmoid', False, 'kdin_query'),
                                                              ('narrow', 'softmax', True, 'kdin_query_small')])
def test_keras_din_restatement(tag, normalizer, need_target, query):
  """layers/keras/din.py DIN: the attention MLP forced to (no final BatchNorm, final bias, linear), softmax or
  sigmoid(score / sqrt(E)) normaliser, a target narrower than the sequence embedding zero-padded for the scores and
  the keys cut back for the output, the target appended when need_target_feature."""
  from google.protobuf import text_format

  from easyrec_amd.protos import seq_encoder_pb2
  cfg = seq_encoder_pb2.DINEncoder()
  text_format.Merge("attention_dnn { hidden_units: [6, 1] activation: 'relu' } attention_normalizer: '%s' "
                    'need_target_feature: %s' % (normalizer, 'true' if need_target else 'false'), cfg)
  state = {k[len('var:kdin_%s/' % tag):]: G[k] for k in G.files if k.startswith('var:kdin_%s/' % tag)}
  state = {'din/din_attention/' + k: v for k, v in state.items()}
  for k in list(state):
    if k.endswith('/bn/gamma'):
      n = state[k].shape[0]
      state[k[:-len('gamma')] + 'moving_mean'], state[k[:-len('gamma')] + 'moving_variance'] = np.zeros(n), np.ones(n)
  got = _oracle()._keras_din(_vars(state), _t('kdin_keys'), torch.from_numpy(np.asarray(G['kdin_len'])), _t(query), cfg, 0.0)
  _close(got, _t('kdin_%s_out' % tag))


@pytest.mark.parametrize('tag,key,kw', [('with_key', 'ta_key', {}), ('no_key', 'ta_key', {'need_key_feature': False}),
                                        ('narrow_key', 'ta_key_narrow', {'allow_key_transform': True})])
def test_product_target_attention_on_the_stand_in_backend(ref_backend, tag, key, kw):
  """THE PRODUCT's layers/sequence_feature_layer.target_attention (sequence_features of a feature group) on the oracle's
  stand-in backend, fed the variables the reference's SequenceFeatureLayer.target_attention created, against that
  function's output: with / without the key in the output, a key narrower than the history zero-padded
  (allow_key_transform) and appended in its padded form."""
  from easyrec_amd.core import context
  from easyrec_amd.core.variables import VarStore
  from easyrec_amd.layers.sequence_feature_layer import target_attention
  from easyrec_amd.protos import dnn_pb2
  cfg = dnn_pb2.DNN()
  cfg.hidden_units.extend([6, 3, 1])
  name = 'ta_' + tag
  fea = {'key': _t(key, torch.float32), 'hist_seq_emb': _t('ta_hist', torch.float32),
         'hist_seq_len': torch.from_numpy(np.asarray(G['din_len'])).to(torch.int32), 'aux_hist_seq_emb_list': []}
  vs = VarStore('cpu', seed=0)
  ctx = context.ModelContext(vs, None, is_training=True)
  with context.use(ctx), torch.no_grad():
    target_attention(cfg, fea, name, None, True, **kw)  # build pass: creates the variables
    ctx.building = False
    state = {k[len('var:'):]: np.asarray(G[k], dtype=np.float32) for k in G.files if k.startswith('var:' + name + '/')}
    vs.load_state_dict(state, strict=False)
    got = target_attention(cfg, fea, name, None, True, **kw)
  _close(got, _t('ta_%s_out' % tag), 2e-5)


def _product(fn, var_prefixes, rename=None):
  """Run a PRODUCT layer (host code of easyrec_amd/layers on the stand-in kernels) with the variables the reference's code
  created: build pass (creates the variables under the product's names), load the recorded values, run."""
  from easyrec_amd.core import context
  from easyrec_amd.core.variables import VarStore
  vs = VarStore('cpu', seed=0)
  ctx = context.ModelContext(vs, None, is_training=True)
  with context.use(ctx), torch.no_grad():
    fn()
    ctx.building = False
    state = {}
    for k in G.files:
      if k.startswith('var:') and any(k[4:].startswith(p) for p in var_prefixes):
        name = k[4:]
        state[rename(name) if rename else name] = np.asarray(G[k], dtype=np.float32)
    missing = [n for n in vs.trainable_names() if n not in state]
    assert not missing, 'the product creates variables the reference did not: %s' % missing
    vs.load_state_dict(state, strict=False)
    return fn()


class _P(object):  # keras-layer parameters from keyword arguments (layers/utils.py Parameter's accessors)
  l2_regularizer = None

  def __init__(self, **kw):
    self.kw = kw

  def get_or_default(self, k, d):
    return self.kw.get(k, d)

  def check_required(self, k):
    assert k in self.kw

  def __getattr__(self, k):
    try:
      return self.__dict__['kw'][k]
    except KeyError:
      raise AttributeError(k)


def test_product_layers_on_the_stand_in_backend(ref_backend):
  """easyrec_amd's own DNN, MMOE, keras MLP / Cross / CIN layers, fed the reference's variables BY NAME (so the variable
  naming - the checkpoint-compatibility contract - is checked too), against the reference code's outputs."""
  from easyrec_amd.layers import dnn, mmoe
  from easyrec_amd.layers.keras import CIN, MLP, Cross
  from easyrec_amd.protos import dnn_pb2

  def dnn_cfg(units, act=None):
    c = dnn_pb2.DNN()
    c.hidden_units.extend(units)
    if act:
      c.activation = act
    return c

  x = _t('dnn_x', torch.float32)
  _close(_product(lambda: dnn.DNN(dnn_cfg([6, 3]), None, 'tower', True)(x), ['tower/']), _t('dnn_out'), 2e-5)
  _close(_product(lambda: dnn.DNN(dnn_cfg([6, 3]), None, 'tower2', True, last_layer_no_activation=True,
                                  last_layer_no_batch_norm=True)(x), ['tower2/']), _t('dnn_out_last_plain'), 2e-5)
  xm = _t('mmoe_x', torch.float32)
  tasks = _product(lambda: mmoe.MMOE(dnn_cfg([5, 3], 'relu'), None, num_task=2, num_expert=3, name='mmoe',
                                     is_training=True)(xm), ['mmoe/'])
  _close(tasks[0], _t('mmoe_task_0'), 2e-5)
  _close(tasks[1], _t('mmoe_task_1'), 2e-5)
  xl = _t('mlp_x', torch.float32)
  from google.protobuf import struct_pb2, text_format

  from easyrec_amd.layers.utils import Parameter
  from easyrec_amd.protos import dnn_pb2 as _dnn_pb2
  for tag, text in (('default', 'hidden_units: [6, 3]'),
                    ('final_linear', "hidden_units: [4, 1] use_final_bn: false final_activation: 'linear'"),
                    ('biased', 'hidden_units: [4, 2] use_bias: true use_final_bias: true use_bn: false'),
                    ('final_none', "hidden_units: [5, 2] final_activation: ''")):
    pb = _dnn_pb2.MLP()
    text_format.Merge(text, pb)
    got = _product(lambda: MLP(Parameter.make_from_pb(pb), name='mlp_%s' % tag)(xl, training=True), ['mlp_%s/' % tag])
    _close(got, _t('mlp_%s_out' % tag), 2e-5)
  # the same block through st_params: an unset final_activation is the layer's default (none) there
  st = struct_pb2.Struct()
  st.update({'hidden_units': [6, 3]})
  got = _product(lambda: MLP(Parameter(st, True), name='mlp_struct')(xl, training=True), ['mlp_struct/'])
  _close(got, _t('mlp_struct_out'), 2e-5)
  assert float((_t('mlp_struct_out') < 0).sum()) > 0 and float((_t('mlp_default_out') < 0).sum()) == 0
  x0, xx = _t('cross_x0', torch.float32), _t('cross_x', torch.float32)
  for tag, kw in (('full', {}), ('diag', {'diag_scale': 0.25})):
    got = _product(lambda: Cross(_P(**kw), name='cross')((x0, xx)), ['cross_%s_' % tag],
                   rename=lambda n: {'cross_%s_kernel' % tag: 'cross/dense/kernel', 'cross_%s_bias' % tag: 'cross/dense/bias'}[n])
    _close(got, _t('cross_%s_out' % tag), 2e-5)
  got = _product(lambda: CIN(_P(hidden_feature_sizes=[5, 2]), name='cin')(_t('cin_x', torch.float32)), ['cin_kernel', 'cin_bias'],
                 rename=lambda n: 'cin/' + n)
  _close(got, _t('cin_out'), 2e-5)


def test_dice_and_gelu(ref_backend):
  """utils/activation.py dice (the DNN's `activation: "dice"`: statistics-only BatchNorm with epsilon 1e-9 over every
  axis but the last, sigmoid gate, alpha under `alpha_<name>`) and gelu: the oracle's restatement and THE PRODUCT's
  easyrec_amd/utils/activation.py (on the stand-in kernel er_dice_fwd is restated by)."""
  from easyrec_amd.utils import activation
  x = _t('act_x')
  _close(_oracle().dice(_layer_vars(), x, 'tower/dnn_0/act'), _t('dice_out'), 1e-8)
  x32 = x.float()
  for name, shape, want in (('tower/dnn_0/act', (11, 6), 'dice_out'), ('att/dnn_1/act', (11, 2, 3), 'dice_3d_out')):
    got = _product(lambda: activation.get_activation('dice', training=True)(x32.reshape(shape), name=name),
                   ['alpha_' + name])
    _close(got, _t(want), 2e-5)
  _close(activation.gelu(x32), _t('gelu_out'), 1e-5)


# ------------------------------------------------------------------------------------------------ model assemblies
import sys  # noqa: E402
sys.path.insert(0, os.path.join(HERE, 'golden'))
import model_assembly_cases as mac  # noqa: E402


class _Groups(object):
  """stands in for the product's InputLayer / SeqInputLayer: `layer(features, group)` -> (concatenation, feature list)
  or the DIN tower's dict, from the fixture"""

  def __init__(self, tag, groups):
    self.data = {}
    for gname, spec in groups.items():
      if spec[0] == 'cat':
        feats = [_t('m:%s:%s:%d' % (tag, gname, i), torch.float32) for i in range(len(spec[1]))]
        self.data[gname] = (torch.cat(feats, dim=1), feats)
      else:
        self.data[gname] = {'key': _t('m:%s:%s:key' % (tag, gname), torch.float32),
                            'hist_seq_emb': _t('m:%s:%s:hist_seq_emb' % (tag, gname), torch.float32),
                            'hist_seq_len': torch.from_numpy(np.asarray(G['m:%s:%s:hist_seq_len' % (tag, gname)])).to(torch.int32),
                            'aux_hist_seq_emb_list': []}

  def has_group(self, name):
    return name in self.data

  def __call__(self, features, group, **kwargs):
    return self.data[group]


@pytest.mark.parametrize('tag', list(mac.CASES))
def test_product_model_assemblies(ref_backend, tag):
  """build_predict_graph of THE PRODUCT's model classes (easyrec_amd/model/*.py, host code over the stand-in kernels)
  against build_predict_graph of the reference's model classes (run by the generator on the same group features): the
  order of the concatenations, which DNN gets which input, variable names (the reference's variables are loaded BY
  NAME) - and the modes: the experts of MMoE / DBMTL normalise with the moving statistics, as the reference's do."""
  from easyrec_amd.model import (dbmtl, dcn, deepfm, dlrm, fm, mmoe, multi_task_model, multi_tower, multi_tower_din,
                                 ple, simple_multi_task, wide_and_deep)
  model, text, groups = mac.CASES[tag]
  cls = {'deepfm': deepfm.DeepFM, 'fm': fm.FM, 'dcn': dcn.DCN, 'wide_and_deep': wide_and_deep.WideAndDeep,
         'dlrm': dlrm.DLRM, 'multi_tower': multi_tower.MultiTower, 'multi_tower_din': multi_tower_din.MultiTowerDIN,
         'simple_multi_task': simple_multi_task.SimpleMultiTask, 'mmoe': mmoe.MMoE, 'ple': ple.PLE,
         'dbmtl': dbmtl.DBMTL, 'multi_task_model': multi_task_model.MultiTaskModel}[model]
  cfg = mac.sub_config(model, text)
  layer = _Groups(tag, groups)
  got = {}
  if model == 'multi_task_model':  # towers over a backbone: its output(s) come from the fixture

    def backbone(self):
      return layer.data['backbone'][0] if 'backbone' in layer.data else [layer.data[n][0] for n in groups]

    # (model classes register themselves by name: one name per case)
    cls = type(cls)('MultiTaskOver_' + tag, (cls,), {'has_backbone': True, 'backbone': property(backbone)})

  def run():
    obj = object.__new__(cls)  # (the constructor builds the input layer, which is not what is compared here)
    obj._model_config, obj._l2_reg, obj._is_training, obj._num_class = cfg, None, True, 1
    obj._prediction_dict, obj._feature_dict, obj._input_layer, obj._seq_input_layer = {}, None, layer, layer
    obj._labels, obj._label_name_dict = None, {}
    if model in ('deepfm', 'fm', 'wide_and_deep'):
      obj._wide_output_dim = cfg.wide_output_dim if model == 'wide_and_deep' else 1
    if model in ('multi_tower', 'multi_tower_din'):
      obj._tower_num, obj._din_tower_num = len(cfg.towers), len(cfg.din_towers)
    obj._outputs, obj._towers = [], []
    if hasattr(cfg, 'task_towers'):
      obj._init_towers(cfg.task_towers)
    obj._add_to_prediction_dict = lambda o: got.__setitem__('out', o)
    obj.build_predict_graph()
    return got['out']

  prefix = tag + '::'
  out = _product(run, [prefix], rename=lambda n: n[len(prefix):])
  if isinstance(out, dict):
    want = {k[len('m:%s:out:' % tag):]: k for k in G.files if k.startswith('m:%s:out:' % tag)}
    assert sorted(out) == sorted(want)
    for name, key in want.items():
      _close(out[name], _t(key), 5e-5)
  else:
    _close(out, _t('m:%s:out' % tag), 5e-5)


@pytest.mark.parametrize('tag,text', [('mlp', 'num_task: 2 num_expert: 3 expert_mlp { hidden_units: [5, 3] }'),
                                      ('plain_final', "num_task: 3 num_expert: 2 expert_mlp { hidden_units: [4] use_final_bn: false "
                                       "final_activation: 'linear' use_final_bias: true }")])
def test_product_keras_mmoe(ref_backend, tag, text):
  """easyrec_amd's keras MMoE block, configured by ITS Parameter over the layer_pb2 message, against the reference's
  layers/keras/multi_task.py MMoE configured by the reference's Parameter over the same message (what an unset
  `final_activation` means - None, not the proto default 'relu' - comes out of Parameter.get_or_default)."""
  from google.protobuf import text_format

  from easyrec_amd.layers.keras import MMoE
  from easyrec_amd.layers.utils import Parameter
  from easyrec_amd.protos import layer_pb2
  pb = layer_pb2.MMoELayer()
  text_format.Merge(text, pb)
  name = 'kmmoe_' + tag
  tasks = _product(lambda: MMoE(Parameter.make_from_pb(pb), name=name)(_t('kmmoe_x', torch.float32), training=True),
                   [name + '/'])
  assert len(tasks) == pb.num_task
  for t, got in enumerate(tasks):
    _close(got, _t('kmmoe_%s_task_%d' % (tag, t)), 2e-5)


@pytest.mark.parametrize('tag,text', [('default', 'reduction_ratio: 4'),
                                      ('bare', 'reduction_ratio: 2 num_squeeze_group: 1 use_skip_connection: false '
                                       'use_output_layer_norm: false')])
def test_product_senet(ref_backend, tag, text):
  """the SENet block in front of MMoE in the reference's mmoe_backbone_on_taobao.config (layers/keras/fibinet.py)"""
  from google.protobuf import text_format

  from easyrec_amd.layers.keras import SENet
  from easyrec_amd.layers.utils import Parameter
  from easyrec_amd.protos import layer_pb2
  pb = layer_pb2.SENet()
  text_format.Merge(text, pb)
  name = 'senet_' + tag
  fields = [_t('senet_in_%d' % i, torch.float32) for i in range(3)]
  got = _product(lambda: SENet(Parameter.make_from_pb(pb), name=name)(fields, training=True), [name + '/'])
  _close(got, _t('senet_%s_out' % tag), 2e-5)


import backbone_cases as bbc  # noqa: E402


class _BackboneGroups(object):
  """stands in for the product's InputLayer under a backbone: `layer(features, group, is_combine)`"""

  def __init__(self, tag, groups):
    self.data = {}
    for gname, spec in groups.items():
      if spec[0] == 'cat':
        self.data[gname] = [_t('b:%s:%s:%d' % (tag, gname, i), torch.float32) for i in range(len(spec[1]))]
      else:
        self.data[gname] = {'seq': _t('b:%s:%s:seq' % (tag, gname), torch.float32),
                            'len': torch.from_numpy(np.asarray(G['b:%s:%s:len' % (tag, gname)])).to(torch.int32),
                            'targets': [_t('b:%s:%s:target:%d' % (tag, gname, i), torch.float32) for i in range(len(spec[3]))]}

  def has_group(self, name):
    return name in self.data

  def __call__(self, features, group, is_combine=True):
    d = self.data[group]
    if isinstance(d, dict):
      assert not is_combine
      return [(d['seq'], d['len'])], None, list(d['targets'])
    return torch.cat(d, dim=1), list(d)


def _lowrank_cross_names(name):
  # keras numbers the unnamed Dense sub-layers of a low-rank Cross (dense, dense_1 per graph: no stable names); this
  # package calls them dense_u / dense_v and keeps the bias under `dense`
  import re
  m = re.match(r'(cross_\d+)/(dense|dense_1)/(kernel|bias)$', name)
  if not m:
    return name
  layer, sub, what = m.groups()
  if what == 'bias':
    return '%s/dense/bias' % layer
  return '%s/%s/kernel' % (layer, 'dense_u' if sub == 'dense' else 'dense_v')


@pytest.mark.parametrize('tag', list(bbc.CASES))
def test_product_backbone(ref_backend, tag):
  """THE PRODUCT's layers/backbone.py (Backbone / Package and the keras layers it loads) on the configs of
  tests/golden/backbone_cases.py against the REFERENCE's layers/backbone.py run by the generator on the same features
  and variables: block order, input_fn / input_slice / extra_input_fn / ignore_input, merges, implicit input blocks,
  keras / lambda / recurrent / repeat / sequential layers, concat_blocks or the leaves, top_mlp, list outputs."""
  from easyrec_amd.layers.backbone import Backbone
  text, groups = bbc.CASES[tag]
  cfg = bbc.backbone_config(text)
  layer = _BackboneGroups(tag, groups)
  prefix = tag + '::'
  rename = (lambda n: _lowrank_cross_names(n[len(prefix):])) if tag == 'bb_dcn_v2_lowrank' else (lambda n: n[len(prefix):])
  out = _product(lambda: Backbone(cfg, None, layer, l2_reg=None)(True), [prefix], rename=rename)
  if isinstance(out, (list, tuple)):
    assert len(out) == len([k for k in G.files if k.startswith('b:%s:out:' % tag)])
    for i, o in enumerate(out):
      _close(o, _t('b:%s:out:%d' % (tag, i)), 5e-5)
  else:
    _close(out, _t('b:%s:out' % tag), 5e-5)


# ------------------------------------------------------------------------------------------------ the HIP kernels
@pytest.mark.gpu
def test_hip_kernels_against_the_reference_layers():
  from easyrec_amd import kernels
  hip, dev = kernels.hip(), 'cuda:0'
  x = _t('fm_inputs', torch.float32).to(dev)
  B, F, D = x.shape
  fm = kernels.FMFn.apply(x.reshape(B, F * D).contiguous(), F, D)
  _close(fm, _t('fm_layers_fm'), 1e-5)
  _close(kernels.RowSumFn.apply(fm), _t('fm_keras'), 1e-5)
  from oracle.kernel_ref import RefBackend
  for si in (False, True):
    keras, where = _t('dot_self%d_skip0' % int(si)), _keras_dot_index(F, si)
    want = torch.stack([keras[:, where[(j, i)]] for i, j in RefBackend._dot_pairs(F, si)], dim=1)
    _close(hip.dot_interaction_fwd(x.reshape(B, F * D).contiguous(), F, D, si), want, 1e-5)
  x0, xx = _t('cross_x0', torch.float32).to(dev), _t('cross_x', torch.float32).to(dev)
  w, b = _t('cross_diag_kernel', torch.float32).to(dev), _t('cross_diag_bias', torch.float32).to(dev)
  u = hip.gemm(kernels.GEMM_NN, xx, w)
  _close(hip.cross_v2_fwd(x0, xx, u, b, 0.25), _t('cross_diag_out'), 1e-5)
  cx = _t('cin_x', torch.float32).to(dev)
  ws = [_t('cin_kernel_%d' % i, torch.float32).to(dev) for i in range(2)]
  bs = [_t('cin_bias_%d' % i, torch.float32).to(dev) for i in range(2)]
  _close(kernels.CINFn.apply(cx, 2, *ws, *bs, None, None, None, None), _t('cin_out'), 1e-5)
