#!/usr/bin/env python
"""Golden vectors from the REFERENCE'S OWN layer code (run in the build container, where /root/reference exists).

TensorFlow cannot be installed here, but several of the reference's layers are a few lines of plain tensor algebra:
  easy_rec/python/layers/fm.py                 FM.__call__            (DeepFM's pairwise term)
  easy_rec/python/layers/keras/interaction.py  FM.call, DotInteraction.call, Cross.call (DCN-v2), CIN.call (xDeepFM)
  easy_rec/python/core/learning_schedules.py   exponential_decay_with_burnin
  easy_rec/python/layers/dnn.py                DNN.__call__           (dense -> batch_normalization -> activation)
  easy_rec/python/model/multi_tower_din.py     MultiTowerDIN.din      (target attention over a padded history)
  easy_rec/python/layers/mmoe.py               MMOE.__call__          (experts, softmax gates, mixture per task)
  easy_rec/python/model/dcn.py                 DCN._cross_net         (DCN-v1 cross layers)
  easy_rec/python/layers/keras/blocks.py       MLP.__init__ / call    (which sub-layers a backbone MLP consists of)
  easy_rec/python/layers/keras/din.py          DIN.__init__ / call    (target attention block: softmax and sigmoid)
  easy_rec/python/layers/sequence_feature_layer.py  SequenceFeatureLayer.target_attention (with / without the key,
                                                a key narrower than the history under allow_key_transform)
  easy_rec/python/layers/keras/multi_task.py   MMoE.call             (expert MLPs or a list of expert inputs, softmax gates)
  easy_rec/python/layers/keras/fibinet.py      SENet.call            (squeeze max / mean per group, excite, re-weight)
  easy_rec/python/layers/backbone.py           Backbone / Package    (the block DAG: input layers, input_fn / input_slice /
                                                extra_input_fn, merges, keras / lambda / recurrent / repeat / sequential
                                                layers, concat_blocks or leaves, top_mlp) with utils/dag.py and
                                                layers/common_layers.py EnhancedInputLayer: tests/golden/backbone_cases.py
  easy_rec/python/model/{deepfm,fm,dcn,wide_and_deep,dlrm,multi_tower,multi_tower_din,simple_multi_task,mmoe,ple,dbmtl}.py
                                                build_predict_graph   (the model classes' assembly of those layers, on
                                                seeded group features: tests/golden/model_assembly_cases.py)
This script executes THOSE FUNCTIONS, unmodified, against a small stand-in for the `tensorflow` module (numpy, fp64)
that implements the documented semantics of the ~25 ops they call (stack, reduce_sum, matmul(transpose_b), band_part,
boolean_mask, tile, sequence_mask, ...), a `keras.layers.Dense` whose kernel / bias are set by this script, and
`tf.layers.dense` / `tf.layers.batch_normalization` (training mode: batch statistics over all axes but the last, epsilon
1e-3, the documented defaults) over seeded variables recorded under their TF names, and stores seeded inputs, those
variables and the outputs in tests/golden/reference_layer_vectors.npz.  tests/test_reference_layers.py then holds the oracle's
restatements (and, on the GPU, the HIP kernels) to those outputs: the index conventions of the reference code (which
axis is summed, kernel[n, h, m] against x_k[h] and x_0[m], lower-triangle order of the dot interaction, diag_scale
placement, staircase flooring) are pinned by the reference itself rather than by a reading of it.

usage: python tests/golden/make_reference_layer_vectors.py [/root/reference]
"""
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------ the tf stand-in
def _arr(x):
  return np.asarray(x, dtype=np.float64) if not isinstance(x, np.ndarray) or x.dtype != np.bool_ else x


class _Shape(tuple):
  """tf.TensorShape's `as_list()` / `ndims` on top of a numpy shape"""

  def as_list(self):
    return list(self)

  @property
  def ndims(self):
    return len(self)


class _Tensor(np.ndarray):
  """a numpy array whose `.shape` also answers TensorShape calls (the reference reads `keys.shape.as_list()`)"""

  @property
  def shape(self):
    return _Shape(np.ndarray.shape.__get__(self))

  def get_shape(self):
    return self.shape


def _tensor(x):
  return np.asarray(x, dtype=np.float64).view(_Tensor)


NEST = [False]  # the later sections switch the nested naming on (the earlier ones prefix by hand through Dense.scope)
SCOPES = []
UNNAMED = {}


def _nested(name, own_call):
  """variable prefix of a stand-in keras layer: Dense.scope + the enclosing layers' names (+ its own)"""
  outer = SCOPES[:-1] if (NEST[0] and own_call) else SCOPES
  return Dense.scope + ''.join(n + '/' for n in outer) + name


class _Layer(object):

  def __init__(self, name=None, **kwargs):
    if name is None and self.__class__.__name__ in ('Dense', 'Add'):
      # keras numbers unnamed layers per graph (dense, dense_1, ...): no stable name to pin; here per enclosing layer
      base = self.__class__.__name__.lower()
      n = UNNAMED[(MODEL_SCOPE[0],) + tuple(SCOPES), base] = UNNAMED.get((((MODEL_SCOPE[0],) + tuple(SCOPES)), base), -1) + 1
      name = base if n == 0 else '%s_%d' % (base, n)
    self.name = self._name = name
    self.built = False
    self.dtype = 'float64'

  def build(self, input_shape):
    self.built = True

  def __call__(self, inputs, *args, **kwargs):  # (keras passes `training` positionally: din.py:45)
    if not NEST[0]:
      return self.call(inputs, *args, **kwargs)
    # keras: a layer called inside another layer's call() builds under both name scopes (outer/inner/kernel)
    SCOPES.append(self.name)
    try:
      if not self.built:
        self.build([_tensor(i).shape for i in inputs] if isinstance(inputs, (list, tuple)) else _tensor(inputs).shape)
      return self.call(inputs, *args, **kwargs)
    finally:
      SCOPES.pop()


class _Dense(_Layer):
  """keras Dense: y = activation(x @ kernel + bias); this script assigns `kernel` / `bias` before the first call."""

  def __init__(self, units, use_bias=True, activation=None, **kwargs):
    super(_Dense, self).__init__(name=kwargs.get('name'))
    self.units, self.use_bias, self.activation = int(units), use_bias, activation  # (keras: int(units))
    self.kernel = self.bias = None

  def call(self, x, **kwargs):
    y = _arr(x) @ self.kernel
    if self.use_bias:
      y = y + self.bias
    return self.activation(y) if self.activation is not None else y



class Dense(_Layer):  # (class names matter: blocks.MLP.call dispatches on layer.__class__.__name__)
  """keras Dense over seeded variables recorded under <scope>/<name>/kernel|bias"""
  scope = ''

  def __init__(self, units, use_bias=True, name=None, activation=None, **kwargs):
    super(Dense, self).__init__(name=name)
    self.units, self.use_bias, self.activation = int(units), use_bias, activation  # (keras: int(units))

  def call(self, x, **kwargs):
    x = _arr(x)
    full = _nested(self.name, True)
    y = x @ VARS.setdefault(full + '/kernel', _VAR_RNG.standard_normal((x.shape[-1], self.units)) * 0.4)
    if self.use_bias:
      y = y + VARS.setdefault(full + '/bias', _VAR_RNG.standard_normal(self.units) * 0.1)
    if self.activation == 'softmax':
      return _softmax(y, axis=-1)
    if self.activation == 'relu':
      return np.maximum(y, 0.0)
    if self.activation == 'sigmoid':
      return 1.0 / (1.0 + np.exp(-y))
    if self.activation == 'tanh':
      return np.tanh(y)
    assert self.activation in (None, 'linear'), self.activation
    return y


class BatchNormalization(_Layer):

  def __init__(self, name=None, trainable=True, **kwargs):
    super(BatchNormalization, self).__init__(name=name)
    self.updates = []

  def __call__(self, x, training=None, **kwargs):
    return _layers_batch_normalization(x, training=training, name=_nested(self.name, False), prescoped=True)


class LayerNormalization(_Layer):
  """keras LayerNormalization defaults: over the last axis, epsilon 1e-3, gamma / beta"""

  def __call__(self, x, **kwargs):
    x = _arr(x)
    full = _nested(self.name, False)
    gamma = VARS.setdefault(full + '/gamma', _VAR_RNG.random(x.shape[-1]) + 0.5)
    beta = VARS.setdefault(full + '/beta', _VAR_RNG.standard_normal(x.shape[-1]) * 0.2)
    mean, var = x.mean(axis=-1, keepdims=True), x.var(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + 1e-3) * gamma + beta


class Add(_Layer):
  """keras.layers.Add: the sum of a list of tensors"""

  def call(self, inputs, **kwargs):
    return sum(_arr(x) for x in inputs)


class Dropout(_Layer):

  def __init__(self, rate, name=None, **kwargs):
    super(Dropout, self).__init__(name=name)

  def __call__(self, x, training=None, **kwargs):
    raise AssertionError('dropout is not exercised')


class _Activation(_Layer):

  def __init__(self, kind, name=None):
    super(_Activation, self).__init__(name=name)
    self.kind = kind

  def call(self, x, **kwargs):
    if self.kind in (None, 'linear'):
      return _arr(x)
    assert self.kind == 'relu', self.kind
    return np.maximum(_arr(x), 0.0)


class _Initializer(object):

  def __init__(self, name='x'):
    self.name = name

  def get_config(self):
    return {'name': self.name}

  @classmethod
  def from_config(cls, cfg):
    return cls(**cfg)


def _reduce_sum(x, axis=None, keepdims=False):
  return np.sum(_arr(x), axis=axis, keepdims=keepdims)


def _band_part(x, num_lower, num_upper):
  x = _arr(x)
  m, n = x.shape[-2], x.shape[-1]
  i, j = np.arange(m)[:, None], np.arange(n)[None, :]
  keep = ((num_lower < 0) | ((i - j) <= num_lower)) & ((num_upper < 0) | ((j - i) <= num_upper))
  return x * keep


VARS = {}  # TF variable name -> value, filled by tf.layers.* below (the consumers feed the same values to the oracle)
_VAR_RNG = np.random.default_rng(77)
VAR_SCOPES = []
MODEL_SCOPE = ['']  # key prefix of the variables a model assembly creates ('<case tag>::'): several cases reuse TF names


def _layers_dense(inputs, units, kernel_regularizer=None, activation=None, name=None, **kw):
  x = _arr(inputs)
  name = MODEL_SCOPE[0] + name
  k = VARS.setdefault(name + '/kernel', _VAR_RNG.standard_normal((x.shape[-1], units)) * 0.4)
  b = VARS.setdefault(name + '/bias', _VAR_RNG.standard_normal(units) * 0.1)
  y = x @ k + b
  return activation(y) if activation is not None else y


def _layers_batch_normalization(inputs, training=False, trainable=True, name=None, epsilon=1e-3, prescoped=False,
                                center=True, scale=True, axis=-1, **kw):
  x = _arr(inputs)
  if not (center or scale):  # Dice's statistics-only normalisation (utils/activation.py:36-42): no gamma / beta
    assert training and axis == -1
    axes = tuple(range(x.ndim - 1))
    return (x - x.mean(axis=axes)) / np.sqrt(x.var(axis=axes) + epsilon)
  name = name if prescoped else MODEL_SCOPE[0] + name
  gamma = VARS.setdefault(name + '/gamma', _VAR_RNG.random(x.shape[-1]) + 0.5)
  beta = VARS.setdefault(name + '/beta', _VAR_RNG.standard_normal(x.shape[-1]) * 0.2)
  if not training:
    # training=False: the moving statistics normalise (TF initialises them to zeros / ones and, with no update op in
    # the graph, they stay there; seeded values here so that the consumers must read them)
    mm = VARS.setdefault(name + '/moving_mean', _VAR_RNG.standard_normal(x.shape[-1]) * 0.3)
    mv = VARS.setdefault(name + '/moving_variance', _VAR_RNG.random(x.shape[-1]) + 0.5)
    return (x - mm) / np.sqrt(mv + epsilon) * gamma + beta
  axes = tuple(range(x.ndim - 1))
  mean, var = x.mean(axis=axes), x.var(axis=axes)
  return (x - mean) / np.sqrt(var + epsilon) * gamma + beta


def _softmax(x, axis=-1):
  x = _arr(x)
  e = np.exp(x - x.max(axis=axis, keepdims=True))
  return e / e.sum(axis=axis, keepdims=True)


def _sequence_mask(lengths, maxlen=None):
  lengths = np.asarray(lengths)
  maxlen = int(lengths.max()) if maxlen is None else int(maxlen)
  return np.arange(maxlen) < lengths[..., None]


def make_tf():
  tf = types.ModuleType('tensorflow')
  tf.__version__ = '1.15.0'
  tf.float32, tf.int32, tf.bool = np.float64, np.int64, np.bool_  # (fp64 throughout: the consumers compare with tolerances)
  tf.name_scope = lambda *a, **k: contextlib.nullcontext()
  tf.stack = lambda xs, axis=0: np.stack([_arr(x) for x in xs], axis=axis)
  tf.concat = lambda xs, axis=-1: np.concatenate([_arr(x) for x in xs], axis=axis)
  tf.square = lambda x: _arr(x) ** 2
  tf.reduce_sum = _reduce_sum
  tf.subtract = lambda a, b: _arr(a) - _arr(b)
  tf.add = lambda a, b: _arr(a) + _arr(b)
  tf.multiply = lambda a, b: _arr(a) * _arr(b)
  tf.maximum = lambda a, b, name=None: np.maximum(_arr(a), _arr(b))
  tf.less = lambda a, b: np.asarray(a) < np.asarray(b)
  tf.constant = lambda v, *a, **k: np.asarray(v)
  tf.expand_dims = lambda x, axis: np.expand_dims(_arr(x), axis)
  tf.tile = lambda x, multiples: np.tile(_arr(x), multiples)
  tf.reshape = lambda x, shape: np.reshape(_arr(x), [int(s) for s in shape])
  tf.shape = lambda x: np.asarray(_arr(x).shape)
  tf.ones_like = lambda x: np.ones_like(_arr(x))
  tf.zeros_like = lambda x: np.zeros_like(_arr(x))
  tf.matmul = lambda a, b, transpose_b=False: _arr(a) @ (np.swapaxes(_arr(b), -1, -2) if transpose_b else _arr(b))
  tf.where = lambda condition=None, x=None, y=None: np.where(condition, x, y)
  tf.cast = lambda x, dtype: np.asarray(x).astype(dtype)
  tf.boolean_mask = lambda t, mask: _arr(t)[np.asarray(mask).astype(bool)]  # row-major order of the kept elements
  tf.linalg = types.SimpleNamespace(band_part=_band_part)
  tf.nn = types.SimpleNamespace(relu=lambda x, name=None: np.maximum(_arr(x), 0.0), softmax=_softmax,
                                sigmoid=lambda x: 1.0 / (1.0 + np.exp(-_arr(x))))
  tf.layers = types.SimpleNamespace(dense=_layers_dense, batch_normalization=_layers_batch_normalization)
  tf.sequence_mask = lambda lengths, maxlen=None, dtype=None: _sequence_mask(lengths, maxlen)
  tf.pad = lambda x, paddings: np.pad(_arr(x), [tuple(p) for p in paddings])
  tf.transpose = lambda x, perm: np.transpose(_arr(x), perm)
  tf.squeeze = lambda x, axis=None: np.squeeze(_arr(x), axis=tuple(axis) if isinstance(axis, (list, tuple)) else axis)
  tf.math = types.SimpleNamespace(add=lambda a, b: _arr(a) + _arr(b))

  def get_variable(name=None, shape=None, dtype=None, **kw):
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(d) for d in shape)
    full = MODEL_SCOPE[0] + ''.join(v + '/' for v in VAR_SCOPES) + name  # (variable scopes, not name scopes)
    return VARS.setdefault(full, _VAR_RNG.standard_normal(shape) * 0.3)

  @contextlib.contextmanager
  def variable_scope(name, *a, **k):
    VAR_SCOPES.append(name)
    try:
      yield
    finally:
      VAR_SCOPES.pop()

  tf.variable_scope = variable_scope
  tf.AUTO_REUSE = 'auto_reuse'

  tf.get_variable = get_variable
  tf.sigmoid = lambda x: 1.0 / (1.0 + np.exp(-_arr(x)))
  tf.errors = types.SimpleNamespace(InvalidArgumentError=ValueError)

  def exponential_decay(lr, step, decay_steps, decay_rate, staircase=False, name=None):  # tf.train.exponential_decay
    p = np.asarray(step, dtype=np.float64) / float(decay_steps)
    if staircase:
      p = np.floor(p)
    return float(lr) * np.power(float(decay_rate), p)

  tf.train = types.SimpleNamespace(exponential_decay=exponential_decay)
  tf.keras = types.SimpleNamespace(
      layers=types.SimpleNamespace(Layer=_Layer, Dense=_Dense),
      activations=types.SimpleNamespace(get=lambda a: a, serialize=lambda a: a),
      initializers=types.SimpleNamespace(get=lambda n: _Initializer(str(n)), serialize=lambda a: a, Zeros=_Initializer),
      regularizers=types.SimpleNamespace(get=lambda r: None, serialize=lambda a: a))
  tf.initializers = types.SimpleNamespace(he_normal=lambda: _Initializer('he_normal'))
  tf.einsum = lambda eq, *ops: np.einsum(eq, *[_arr(o) for o in ops])
  tf.reduce_max = lambda x, axis=None, keepdims=False: np.max(_arr(x), axis=axis, keepdims=keepdims)
  tf.reduce_mean = lambda x, axis=None, keepdims=False: np.mean(_arr(x), axis=axis, keepdims=keepdims)
  tf.add_n = lambda xs: sum(_arr(x) for x in xs)
  tf.nn.bias_add = lambda x, b: _arr(x) + _arr(b)
  tf.zeros_initializer = lambda: _Initializer('zeros')
  tf.logging = types.SimpleNamespace(info=lambda *a, **k: None, warn=lambda *a, **k: None)

  def as_tensor(fn):  # results answer .shape.as_list() / .get_shape() like tf.Tensor does; op names are dropped
    import inspect
    named = 'name' in inspect.signature(fn).parameters or any(
        p.kind == p.VAR_KEYWORD for p in inspect.signature(fn).parameters.values())

    def wrapped(*a, **k):
      if not named:
        k.pop('name', None)
      r = fn(*a, **k)
      if isinstance(r, np.ndarray) and not isinstance(r, _Tensor):
        r = r.view(_Tensor)
      return r
    return wrapped

  for ns in (tf, tf.nn, tf.layers, tf.math, tf.linalg):
    for k, v in list(vars(ns).items()):
      if isinstance(v, types.FunctionType) and k not in ('name_scope',):
        setattr(ns, k, as_tensor(v))
  tf.compat = types.SimpleNamespace(v1=tf)
  return tf


def load_reference(rel_path, name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel_path))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


class Params(object):
  """what the reference's layers read their settings from (layers/utils.py Parameter.get_or_default)"""

  def __init__(self, **kw):
    self.kw = kw

  def get_or_default(self, key, default):
    return self.kw.get(key, default)


_PARAMETER = []


def load_reference_parameter():
  """the reference's layers/utils.py (Parameter), its tensorflow-internal imports stubbed"""
  if not _PARAMETER:
    for name in ('tensorflow.python.framework', 'tensorflow.python.framework.ops', 'tensorflow.python.framework.sparse_tensor',
                 'tensorflow.python.ops', 'tensorflow.python.ops.variables'):
      sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['tensorflow.python.framework'].ops = sys.modules['tensorflow.python.framework.ops']
    sys.modules['tensorflow.python.framework'].sparse_tensor = sys.modules['tensorflow.python.framework.sparse_tensor']
    sys.modules['tensorflow.python.ops'].variables = sys.modules['tensorflow.python.ops.variables']
    _PARAMETER.append(load_reference('easy_rec/python/layers/utils.py', 'ref_layers_utils'))
    sys.modules['easy_rec.python.layers.utils'].Parameter = _PARAMETER[0].Parameter
  return _PARAMETER[0]


def keras_multi_task_and_senet(out, rng, blocks):
  """layers/keras/multi_task.py MMoE and layers/keras/fibinet.py SENet - the blocks of the reference's
  samples/model_config/mmoe_backbone_on_taobao.config - driven by the reference's own Parameter (layers/utils.py) over
  the real layer_pb2 messages, variables named by the nested keras scopes."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
  from google.protobuf import text_format
  from easyrec_amd import protos
  utils_ref = load_reference_parameter()
  sys.modules['easy_rec.python.layers.keras.blocks'] = blocks
  for name, attrs in (('easy_rec.python.layers.keras.attention', {'Attention': object}),
                      ('easy_rec.python.layers.keras.layer_norm', {'LayerNormalization': LayerNormalization})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
      setattr(m, k, v)
    sys.modules[name] = m
  sys.modules['easy_rec.python.protos.seq_encoder_pb2'] = protos.seq_encoder_pb2
  sys.modules['easy_rec.python.protos'].seq_encoder_pb2 = protos.seq_encoder_pb2
  mt = load_reference('easy_rec/python/layers/keras/multi_task.py', 'ref_keras_multi_task')
  fib = load_reference('easy_rec/python/layers/keras/fibinet.py', 'ref_keras_fibinet')
  NEST[0] = True
  try:
    B = 8
    x = rng.standard_normal((B, 6))
    out['kmmoe_x'] = x
    for tag, text, inputs in (('mlp', 'num_task: 2 num_expert: 3 expert_mlp { hidden_units: [5, 3] }', _tensor(x)),
                              ('plain_final', "num_task: 3 num_expert: 2 expert_mlp { hidden_units: [4] use_final_bn: false "
                               "final_activation: 'linear' use_final_bias: true }", _tensor(x))):
      # (without expert_mlp the reference stacks ALL its inputs - the gates' input too - against num_expert gate
      #  columns, multi_task.py:52-64: a shape error for any input list; not exercised)
      pb = protos.layer_pb2.MMoELayer()
      text_format.Merge(text, pb)
      layer = mt.MMoE(utils_ref.Parameter.make_from_pb(pb), name='kmmoe_' + tag)
      for t, v in enumerate(layer(inputs, training=True)):
        out['kmmoe_%s_task_%d' % (tag, t)] = np.asarray(v)
    fields = [rng.standard_normal((B, d)) for d in (4, 6, 2)]
    for i, f in enumerate(fields):
      out['senet_in_%d' % i] = f
    for tag, text in (('default', 'reduction_ratio: 4'),
                      ('bare', 'reduction_ratio: 2 num_squeeze_group: 1 use_skip_connection: false use_output_layer_norm: false')):
      pb = protos.layer_pb2.SENet()
      text_format.Merge(text, pb)
      layer = fib.SENet(utils_ref.Parameter.make_from_pb(pb), name='senet_' + tag)
      out['senet_%s_out' % tag] = np.asarray(layer([_tensor(f) for f in fields]))
  finally:
    NEST[0] = False
  return mt, fib


def backbones(out, rng, layer_classes):
  """layers/backbone.py Backbone / Package on the configs of backbone_cases.py: the reference's own DAG (utils/dag.py),
  EnhancedInputLayer (layers/common_layers.py), Parameter and keras layers; the feature groups come from a stand-in
  input layer holding seeded features."""
  sys.path.insert(0, HERE)
  import backbone_cases as bc
  from easyrec_amd import protos
  tf = sys.modules['tensorflow']
  load_reference_parameter()
  for name, attrs in (('easy_rec.python.compat.layers', {'layer_norm': None}),
                      ('easy_rec.python.utils.load_class', {'load_keras_layer': lambda n: layer_classes.get(n, (None, False))})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
      setattr(m, k, v)
    sys.modules[name] = m
  sys.modules['easy_rec.python.protos.backbone_pb2'] = protos.backbone_pb2
  sys.modules['easy_rec.python.protos'].backbone_pb2 = protos.backbone_pb2
  sys.modules['easy_rec.python.utils.dag'] = load_reference('easy_rec/python/utils/dag.py', 'easy_rec.python.utils.dag')
  sys.modules['easy_rec.python.layers.common_layers'] = load_reference('easy_rec/python/layers/common_layers.py',
                                                                       'easy_rec.python.layers.common_layers')
  keras_pkg = sys.modules['easy_rec.python.layers.keras']
  keras_pkg.MLP, keras_pkg.EmbeddingLayer = layer_classes['MLP'][0], object
  bb = load_reference('easy_rec/python/layers/backbone.py', 'ref_layers_backbone')
  tf.keras.layers.Dense = Dense  # (sub-layers built inside the reference's layers create their own variables here)

  class Groups(object):  # InputLayer.__call__(features, group, is_combine) of the reference, from seeded features

    def __init__(self, data):
      self.data = data

    def has_group(self, name):
      return name in self.data

    def __call__(self, features, group, is_combine=True):
      d = self.data[group]
      if isinstance(d, dict):
        assert not is_combine
        return [(d['seq'], d['len'])], None, list(d['targets'])
      assert is_combine
      return _tensor(np.concatenate([np.asarray(f) for f in d], axis=1)), list(d)

  NEST[0] = True
  try:
    for tag, (text, groups) in bc.CASES.items():
      data = {}
      for gname, spec in groups.items():
        if spec[0] == 'cat':
          data[gname] = [_tensor(rng.standard_normal((bc.B, w))) for w in spec[1]]
          for i, f in enumerate(data[gname]):
            out['b:%s:%s:%d' % (tag, gname, i)] = np.asarray(f)
        else:
          _, E, L, tw = spec
          lens = rng.integers(1, L + 1, bc.B).astype(np.int64)
          lens[0] = L
          data[gname] = {'seq': _tensor(rng.standard_normal((bc.B, L, E))), 'len': lens,
                         'targets': [_tensor(rng.standard_normal((bc.B, w))) for w in tw]}
          out['b:%s:%s:seq' % (tag, gname)], out['b:%s:%s:len' % (tag, gname)] = np.asarray(data[gname]['seq']), lens
          for i, f in enumerate(data[gname]['targets']):
            out['b:%s:%s:target:%d' % (tag, gname, i)] = np.asarray(f)
      MODEL_SCOPE[0] = Dense.scope = tag + '::'
      try:
        res = bb.Backbone(bc.backbone_config(text), None, Groups(data), l2_reg=None)(True)
      finally:
        MODEL_SCOPE[0] = Dense.scope = ''
      if isinstance(res, (list, tuple)):
        for i, r in enumerate(res):
          out['b:%s:out:%d' % (tag, i)] = np.asarray(r)
      else:
        out['b:%s:out' % tag] = np.asarray(res)
  finally:
    NEST[0] = False


def model_assemblies(out, rng, loaded):
  """build_predict_graph of the reference's model classes, on bare instances (no constructor: the input layer, the
  estimator and the loss are not what is pinned here) holding seeded group features and the model's config message
  (easyrec_amd.protos is the reference's schema: the drop-in boundary)."""
  sys.path.insert(0, HERE)
  sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
  import model_assembly_cases as mac
  from easyrec_amd import protos
  for base in ('deepfm', 'fm', 'dcn', 'wide_and_deep', 'dlrm', 'multi_tower', 'simple_multi_task', 'mmoe', 'ple', 'dbmtl',
               'tower', 'easy_rec_model', 'loss'):
    sys.modules['easy_rec.python.protos.%s_pb2' % base] = getattr(protos, base + '_pb2')
  sys.modules['easy_rec.python.model.multi_task_model'] = types.ModuleType('easy_rec.python.model.multi_task_model')
  sys.modules['easy_rec.python.model.multi_task_model'].MultiTaskModel = object
  for name in ('cmbf', 'uniter'):
    m = types.ModuleType('easy_rec.python.layers.' + name)
    sys.modules[m.__name__] = m
    setattr(sys.modules['easy_rec.python.layers'], name, m)
  sys.modules['easy_rec.python.protos'].tower_pb2 = protos.tower_pb2
  builders = types.ModuleType('easy_rec.python.builders')
  builders.loss_builder = types.ModuleType('easy_rec.python.builders.loss_builder')
  sys.modules['easy_rec.python.builders'], sys.modules['easy_rec.python.builders.loss_builder'] = builders, builders.loss_builder
  # (loaded before easy_rec.python.model.multi_task_model is replaced by the stub the multi-task models derive from)
  mt_mod = load_reference('easy_rec/python/model/multi_task_model.py', 'ref_model_multi_task_model')
  classes = {'multi_task_model': 'MultiTaskModel', 'deepfm': 'DeepFM', 'fm': 'FM', 'dcn': 'DCN', 'wide_and_deep': 'WideAndDeep', 'dlrm': 'DLRM',
             'multi_tower': 'MultiTower', 'multi_tower_din': 'MultiTowerDIN', 'simple_multi_task': 'SimpleMultiTask',
             'mmoe': 'MMoE', 'ple': 'PLE', 'dbmtl': 'DBMTL'}
  mods = dict(loaded)
  mods['multi_task_model'] = mt_mod
  for tag, (model, text, groups) in mac.CASES.items():
    if model not in mods:
      mods[model] = load_reference('easy_rec/python/model/%s.py' % model, 'ref_model_' + model)
    cls = getattr(mods[model], classes[model])
    cfg = mac.sub_config(model, text)
    # seeded group features
    g = {}
    for gname, spec in groups.items():
      if spec[0] == 'cat':
        g[gname] = [_tensor(rng.standard_normal((mac.B, w))) for w in spec[1]]
        for i, f in enumerate(g[gname]):
          out['m:%s:%s:%d' % (tag, gname, i)] = np.asarray(f)
      else:
        _, E, L = spec
        lens = rng.integers(1, L + 1, mac.B).astype(np.int64)
        lens[0] = L  # (tf.sequence_mask(lengths) takes its width from the longest)
        g[gname] = {'key': _tensor(rng.standard_normal((mac.B, E))), 'hist_seq_emb': _tensor(rng.standard_normal((mac.B, L, E))),
                    'hist_seq_len': lens}
        for k, v in g[gname].items():
          out['m:%s:%s:%s' % (tag, gname, k)] = np.asarray(v)
    cat = lambda name: _tensor(np.concatenate([np.asarray(f) for f in g[name]], axis=1))
    got = {}
    obj = object.__new__(cls)
    obj._model_config, obj._l2_reg, obj._is_training, obj._num_class = cfg, None, True, 1
    obj._prediction_dict = {}
    obj._add_to_prediction_dict = lambda o: got.__setitem__('out', o)
    if model in ('deepfm', 'fm'):
      obj._wide_output_dim = 1
      obj._wide_features, obj._deep_features = cat('wide'), cat('deep')
      obj._fm_features = list(g['fm'] if 'fm' in g else g['deep'])
    elif model == 'dcn':
      obj._features = cat('all')
    elif model == 'wide_and_deep':
      obj._wide_features, obj._deep_features = list(g['wide']), list(g['deep'])
    elif model == 'dlrm':
      obj._sparse_features, obj._dense_feature = list(g['sparse']), cat('dense')
    elif model in ('multi_tower', 'multi_tower_din'):
      obj._tower_num = len(cfg.towers)
      obj._tower_features = [cat(t.input) for t in cfg.towers]
      if model == 'multi_tower_din':
        obj._din_tower_num = len(cfg.din_towers)
        obj._din_tower_features = [g[t.input] for t in cfg.din_towers]
    elif model == 'multi_task_model':  # towers over a backbone's output(s)
      obj.has_backbone, obj._outputs, obj._labels, obj._label_name_dict = True, [], None, {}
      obj.backbone = cat('backbone') if 'backbone' in g else [cat(n) for n in g]
    else:  # the multi-task models
      obj._features = cat('all')
      obj._task_towers = cfg.task_towers
      obj._task_num = obj._task_nums = len(cfg.task_towers)
      obj.backbone = None
    MODEL_SCOPE[0] = tag + '::'
    try:
      cls.build_predict_graph(obj)
    finally:
      MODEL_SCOPE[0] = ''
    res = got['out']
    if isinstance(res, dict):
      for k, v in res.items():
        out['m:%s:out:%s' % (tag, k)] = np.asarray(v)
    else:
      out['m:%s:out' % tag] = np.asarray(res)


def main():
  sys.modules['tensorflow'] = make_tf()
  for pkg in ('easy_rec', 'easy_rec.python', 'easy_rec.python.utils'):
    sys.modules.setdefault(pkg, types.ModuleType(pkg))
  act = types.ModuleType('easy_rec.python.utils.activation')
  relu = sys.modules['tensorflow'].nn.relu
  act.get_activation = lambda name, **kw: relu if name in ('relu', 'tf.nn.relu', 'nn.relu') else None
  sys.modules['easy_rec.python.utils.activation'] = act
  fm_mod = load_reference('easy_rec/python/layers/fm.py', 'ref_layers_fm')
  inter = load_reference('easy_rec/python/layers/keras/interaction.py', 'ref_keras_interaction')
  sched = load_reference('easy_rec/python/core/learning_schedules.py', 'ref_learning_schedules')
  dnn_mod = load_reference('easy_rec/python/layers/dnn.py', 'easy_rec.python.layers.dnn')
  layers_pkg = types.ModuleType('easy_rec.python.layers')
  layers_pkg.dnn = dnn_mod
  layers_pkg.seq_input_layer = types.ModuleType('easy_rec.python.layers.seq_input_layer')
  sys.modules['easy_rec.python.layers'] = layers_pkg
  sys.modules['easy_rec.python.layers.dnn'] = dnn_mod
  sys.modules['easy_rec.python.layers.seq_input_layer'] = layers_pkg.seq_input_layer
  for name, attrs in (('easy_rec.python.compat', {}), ('easy_rec.python.compat.regularizers', {}),
                      ('easy_rec.python.model', {}), ('easy_rec.python.model.rank_model', {'RankModel': object}),
                      ('easy_rec.python.protos', {}), ('easy_rec.python.protos.multi_tower_pb2', {'MultiTower': object}),
                      ('easy_rec.python.protos.dcn_pb2', {'DCN': object})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
      setattr(m, k, v)
    sys.modules[name] = m
  sys.modules['easy_rec.python.compat'].regularizers = sys.modules['easy_rec.python.compat.regularizers']
  din_mod = load_reference('easy_rec/python/model/multi_tower_din.py', 'ref_multi_tower_din')
  mmoe_mod = load_reference('easy_rec/python/layers/mmoe.py', 'ref_layers_mmoe')
  dcn_mod = load_reference('easy_rec/python/model/dcn.py', 'ref_model_dcn')
  # keras MLP block: its imports of keras classes / helpers resolve to the stand-ins above
  tf = sys.modules['tensorflow']
  tf.GraphKeys = types.SimpleNamespace(UPDATE_OPS='update_ops')
  tf.keras.layers.BatchNormalization = BatchNormalization
  for name, attrs in (('tensorflow.python', {}), ('tensorflow.python.keras', {}),
                      ('tensorflow.python.keras.initializers', {'Constant': _Initializer}),
                      ('tensorflow.python.keras.layers', {'Dense': Dense, 'Dropout': Dropout, 'Lambda': _Layer, 'Layer': _Layer}),
                      ('easy_rec.python.layers.keras', {}),
                      ('easy_rec.python.layers.keras.activation', {'activation_layer': lambda a, name=None: _Activation(a, name)}),
                      ('easy_rec.python.layers.utils', {'Parameter': object}),
                      ('easy_rec.python.utils.shape_utils', {'pad_or_truncate_sequence': None}),
                      ('easy_rec.python.utils.tf_utils', {'add_elements_to_collection': lambda *a, **k: None})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
      setattr(m, k, v)
    sys.modules[name] = m
  blocks = load_reference('easy_rec/python/layers/keras/blocks.py', 'ref_keras_blocks')

  rng = np.random.default_rng(20240923)
  out = {}
  B, F, D = 7, 5, 4
  feats = [rng.standard_normal((B, D)) for _ in range(F)]
  out['fm_inputs'] = np.stack(feats, axis=1)
  out['fm_layers_fm'] = fm_mod.FM()(feats)                                            # [B, D]
  out['fm_keras'] = inter.FM(Params(), name='fm').call(feats)                          # [B, 1]
  out['fm_keras_variant'] = inter.FM(Params(use_variant=True), name='fm').call(feats)  # [B, D]
  for self_inter in (False, True):
    for skip in (False, True):
      layer = inter.DotInteraction(Params(self_interaction=self_inter, skip_gather=skip), name='dot')
      out['dot_self%d_skip%d' % (self_inter, skip)] = layer.call(feats)

  d = 6
  x0, x = rng.standard_normal((B, d)), rng.standard_normal((B, d))
  out['cross_x0'], out['cross_x'] = x0, x
  for tag, kw in (('full', {}), ('diag', {'diag_scale': 0.25}), ('nobias', {'use_bias': False}),
                  ('lowrank', {'projection_dim': 3})):
    layer = inter.Cross(Params(**kw), name='cross')
    layer.build((x0.shape, x.shape))
    layer.built = True
    if 'projection_dim' in kw:
      layer._dense_u.kernel = rng.standard_normal((d, 3)) * 0.5
      layer._dense_v.kernel = rng.standard_normal((3, d)) * 0.5
      layer._dense_v.bias = rng.standard_normal(d) * 0.1
      out['cross_%s_u' % tag], out['cross_%s_v' % tag], out['cross_%s_bias' % tag] = \
          layer._dense_u.kernel, layer._dense_v.kernel, layer._dense_v.bias
    else:
      layer._dense.kernel = rng.standard_normal((d, d)) * 0.5
      layer._dense.bias = rng.standard_normal(d) * 0.1 if kw.get('use_bias', True) else None
      out['cross_%s_kernel' % tag] = layer._dense.kernel
      if layer._dense.bias is not None:
        out['cross_%s_bias' % tag] = layer._dense.bias
    out['cross_%s_out' % tag] = layer.call((x0, x))

  H0, Dc, sizes = 4, 3, [5, 2]
  cin_x = rng.standard_normal((B, H0, Dc))
  cin = inter.CIN(Params(hidden_feature_sizes=sizes), name='cin')
  hs = [H0] + sizes
  cin.kernel_list = [rng.standard_normal((hs[i + 1], hs[i], H0)) * 0.4 for i in range(len(sizes))]
  cin.bias_list = [rng.standard_normal(hs[i + 1]) * 0.2 for i in range(len(sizes))]
  out['cin_x'] = cin_x
  for i in range(len(sizes)):
    out['cin_kernel_%d' % i], out['cin_bias_%d' % i] = cin.kernel_list[i], cin.bias_list[i]
  out['cin_out'] = cin.call(cin_x)

  steps = np.array([0, 1, 5, 9, 10, 11, 999, 1000, 1001, 2500, 25000, 100000], dtype=np.int64)
  out['lr_steps'] = steps
  for tag, kw in (('plain', dict(learning_rate_base=0.001, learning_rate_decay_steps=1000, learning_rate_decay_factor=0.5,
                                 min_learning_rate=1e-5)),
                  ('burnin', dict(learning_rate_base=0.01, learning_rate_decay_steps=500, learning_rate_decay_factor=0.7,
                                  burnin_learning_rate=0.001, burnin_steps=10, min_learning_rate=1e-6)),
                  ('smooth', dict(learning_rate_base=0.002, learning_rate_decay_steps=300, learning_rate_decay_factor=0.9,
                                  staircase=False))):
    out['lr_%s' % tag] = np.array([float(sched.exponential_decay_with_burnin(int(s), **kw)) for s in steps])
    out['lr_%s_args' % tag] = np.array([kw.get(k, d) for k, d in (('learning_rate_base', 0), ('learning_rate_decay_steps', 0),
                                                                 ('learning_rate_decay_factor', 0), ('burnin_learning_rate', 0.0),
                                                                 ('burnin_steps', 0), ('min_learning_rate', 0.0),
                                                                 ('staircase', True))], dtype=np.float64)

  # DNN / DIN / MMoE: tf.layers.dense + batch_normalization over seeded variables recorded in VARS
  dnn_cfg = types.SimpleNamespace(hidden_units=[6, 3], use_bn=True, activation='tf.nn.relu', dropout_ratio=[])
  x_dnn = rng.standard_normal((9, 5))
  out['dnn_x'] = x_dnn
  out['dnn_out'] = dnn_mod.DNN(dnn_cfg, None, 'tower', True)(x_dnn)
  out['dnn_out_last_plain'] = dnn_mod.DNN(dnn_cfg, None, 'tower2', True, last_layer_no_activation=True,
                                          last_layer_no_batch_norm=True)(x_dnn)
  Bd, L, E = 6, 5, 4
  din_cfg = types.SimpleNamespace(hidden_units=[8, 4, 1], use_bn=True, activation='tf.nn.relu', dropout_ratio=[])
  key, hist = rng.standard_normal((Bd, E)), rng.standard_normal((Bd, L, E))
  lens = np.array([5, 1, 3, 2, 5, 4], dtype=np.int64)
  out['din_key'], out['din_hist'], out['din_len'] = key, hist, lens
  fake_self = types.SimpleNamespace(_l2_reg=None, _is_training=True)
  out['din_out'] = din_mod.MultiTowerDIN.din(fake_self, din_cfg, {'key': key, 'hist_seq_emb': hist, 'hist_seq_len': lens},
                                             'din')
  exp_cfg = types.SimpleNamespace(hidden_units=[5, 3], use_bn=True, activation='relu', dropout_ratio=[])
  x_mm = rng.standard_normal((8, 6))
  out['mmoe_x'] = x_mm
  tasks = mmoe_mod.MMOE(exp_cfg, None, num_task=2, num_expert=3, name='mmoe', is_training=True)(x_mm)
  out['mmoe_task_0'], out['mmoe_task_1'] = tasks
  # keras MLP (backbone blocks): default settings, and the settings of xDeepFM's final block - configured as a backbone
  # block is: the reference's own Parameter (layers/utils.py) over the real dnn_pb2.MLP message
  sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
  from google.protobuf import text_format
  from easyrec_amd import protos
  utils_ref = load_reference_parameter()
  x_mlp = rng.standard_normal((9, 5))
  out['mlp_x'] = x_mlp
  for tag, text in (('default', 'hidden_units: [6, 3]'),  # (final_activation unset: 'relu' through Parameter, see below)
                    ('final_linear', "hidden_units: [4, 1] use_final_bn: false final_activation: 'linear'"),
                    ('biased', 'hidden_units: [4, 2] use_bias: true use_final_bias: true use_bn: false'),
                    ('final_none', "hidden_units: [5, 2] final_activation: ''")):
    pb = protos.dnn_pb2.MLP()
    text_format.Merge(text, pb)
    Dense.scope = 'mlp_%s/' % tag
    out['mlp_%s_out' % tag] = blocks.MLP(utils_ref.Parameter.make_from_pb(pb), name='mlp_%s' % tag).call(x_mlp, training=True)
  Dense.scope = ''
  # the same block configured through st_params (a Struct): here an unset final_activation IS the layer's default None
  from google.protobuf import struct_pb2
  st = struct_pb2.Struct()
  st.update({'hidden_units': [6, 3]})
  Dense.scope = 'mlp_struct/'
  out['mlp_struct_out'] = blocks.MLP(utils_ref.Parameter(st, True), name='mlp_struct').call(x_mlp, training=True)
  Dense.scope = ''
  # keras DIN block: its attention MLP is the reference's own blocks.MLP (loaded above) under the name `din_attention`
  keras_pkg = sys.modules['easy_rec.python.layers.keras']
  keras_pkg.MLP = blocks.MLP
  sys.modules['easy_rec.python.utils.shape_utils'].get_shape_list = lambda t, rank=None: list(_arr(t).shape)
  din_keras = load_reference('easy_rec/python/layers/keras/din.py', 'ref_keras_din')
  Bk, Lk, Ek = 6, 5, 4
  keys_k, query_k = rng.standard_normal((Bk, Lk, Ek)), rng.standard_normal((Bk, Ek))
  query_small = rng.standard_normal((Bk, 3))  # a target narrower than the sequence embedding: padded with zeros
  lens_k = np.array([5, 2, 4, 1, 5, 3], dtype=np.int64)
  out['kdin_keys'], out['kdin_query'], out['kdin_query_small'], out['kdin_len'] = keys_k, query_k, query_small, lens_k
  for tag, normalizer, need_target, q in (('softmax', 'softmax', True, query_k), ('sigmoid', 'sigmoid', False, query_k),
                                          ('narrow', 'softmax', True, query_small)):
    pb = protos.seq_encoder_pb2.DINEncoder()
    text_format.Merge("attention_dnn { hidden_units: [6, 1] activation: 'relu' } attention_normalizer: '%s' "
                      'need_target_feature: %s' % (normalizer, 'true' if need_target else 'false'), pb)
    Dense.scope = 'kdin_%s/' % tag
    layer = din_keras.DIN(utils_ref.Parameter.make_from_pb(pb), name='din')
    out['kdin_%s_out' % tag] = np.asarray(layer.call((_tensor(keys_k), lens_k, _tensor(q)), training=True))
  Dense.scope = ''
  # SequenceFeatureLayer.target_attention (sequence_features inside a feature group)
  for name, attrs in (('tensorflow.python.framework', {}), ('tensorflow.python.framework.ops', {}),
                      ('easy_rec.python.utils.conditional', {})):
    m = sys.modules.get(name) or types.ModuleType(name)
    sys.modules[name] = m
  sys.modules['tensorflow.python.framework'].ops = sys.modules['tensorflow.python.framework.ops']
  sys.modules['easy_rec.python.utils'].conditional = sys.modules['easy_rec.python.utils.conditional']
  sfl = load_reference('easy_rec/python/layers/sequence_feature_layer.py', 'ref_sequence_feature_layer')
  fake_layer = types.SimpleNamespace(_kernel_regularizer=None, _is_training=True)
  ta_cfg = types.SimpleNamespace(hidden_units=[6, 3, 1], use_bn=True, activation='tf.nn.relu', dropout_ratio=[])
  ta_key, ta_hist = rng.standard_normal((Bd, E)), rng.standard_normal((Bd, L, E))
  ta_key_narrow = rng.standard_normal((Bd, 3))
  out['ta_key'], out['ta_hist'], out['ta_key_narrow'] = ta_key, ta_hist, ta_key_narrow
  for tag, key, kw in (('with_key', ta_key, {}), ('no_key', ta_key, {'need_key_feature': False}),
                       ('narrow_key', ta_key_narrow, {'allow_key_transform': True})):
    fea = {'key': key, 'hist_seq_emb': ta_hist, 'hist_seq_len': lens, 'aux_hist_seq_emb_list': []}
    out['ta_%s_out' % tag] = sfl.SequenceFeatureLayer.target_attention(fake_layer, ta_cfg, fea, 'ta_' + tag, **kw)
  x_dcn = rng.standard_normal((7, 5))
  out['dcn_x'] = x_dcn
  out['dcn_cross_out'] = dcn_mod.DCN._cross_net(None, x_dcn, 3)
  layers_pkg.fm, layers_pkg.mmoe = fm_mod, mmoe_mod
  model_assemblies(out, np.random.default_rng(20240924), {'dcn': dcn_mod, 'multi_tower_din': din_mod})
  mt, fib = keras_multi_task_and_senet(out, np.random.default_rng(20240925), blocks)
  backbones(out, np.random.default_rng(20240926),
            {'MLP': (blocks.MLP, True), 'Cross': (inter.Cross, True), 'FM': (inter.FM, True), 'CIN': (inter.CIN, True),
             'DotInteraction': (inter.DotInteraction, True), 'DIN': (din_keras.DIN, True), 'MMoE': (mt.MMoE, True),
             'SENet': (fib.SENet, True), 'Add': (Add, False), 'Dense': (Dense, False)})
  # Dice / gelu of utils/activation.py (the DNN's `activation: "dice"`: DIN's attention MLP in the reference's samples)
  sys.modules['easy_rec.python.utils.load_class'] = types.ModuleType('easy_rec.python.utils.load_class')
  sys.modules['easy_rec.python.utils.load_class'].load_by_path = None
  tf.constant_initializer = lambda v: _Initializer('constant')
  tf.tanh = lambda x: np.tanh(_arr(x))
  tf.pow = lambda x, p: np.power(_arr(x), p)
  act_ref = load_reference('easy_rec/python/utils/activation.py', 'ref_utils_activation')
  x_act = rng_act = None
  rng_act = np.random.default_rng(20240929)
  x_act = rng_act.standard_normal((11, 6)) * 1.7 + 0.3
  out['act_x'] = x_act
  out['dice_out'] = np.asarray(act_ref.dice(_tensor(x_act), name='tower/dnn_0/act', training=True))
  out['dice_3d_out'] = np.asarray(act_ref.dice(_tensor(x_act.reshape(11, 2, 3)), name='att/dnn_1/act', training=True))
  out['gelu_out'] = np.asarray(act_ref.gelu(_tensor(x_act)))
  for k, v in VARS.items():
    out['var:' + k] = v
  for k in [k for k in out if k.startswith('cross_') and (k.endswith('_kernel') or k.endswith('_bias'))] + \
      [k for k in out if k.startswith('cin_kernel_') or k.startswith('cin_bias_')]:
    out['var:' + k] = out[k]
  path = os.path.join(HERE, 'reference_layer_vectors.npz')
  np.savez(path, **{k: np.asarray(v) for k, v in out.items()})
  print('wrote %s: %d arrays' % (path, len(out)))


if __name__ == '__main__':
  if os.environ.get('PYTHONHASHSEED') != '0':
    # the reference's DAG keeps a node's successors in a set (utils/dag.py:21): the order in which independent backbone
    # blocks run - hence the order the seeded variables are drawn in - follows the string hash seed.  Fixed for a
    # reproducible fixture (the outputs are compared with variables loaded BY NAME either way).
    os.environ['PYTHONHASHSEED'] = '0'
    os.execv(sys.executable, [sys.executable] + sys.argv)
  main()
