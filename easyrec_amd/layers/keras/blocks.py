"""Keras-style `MLP` block (reference easy_rec/python/layers/keras/blocks.py:22-128).

Differs from the legacy `DNN` (layers/dnn.py): no bias by default, `he_uniform` kernels, BatchNorm +
activation also on the LAST layer unless `use_final_bn` / `final_activation` say otherwise
(SURVEY.md App. B.10).  Variables: `<name>/layer_<i>/dense/kernel[, bias]`, `<name>/layer_<i>/bn/*`.
Per layer: one MFMA GEMM (er_gemm) + one fused bias/BatchNorm/ReLU kernel pair (er_bn_act).
"""
import logging

import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.utils.activation import get_activation, is_relu


class MLP(object):

  def __init__(self, params, name='mlp', reuse=None, **kwargs):
    self.name = name
    self.layer_name = name
    params.check_required('hidden_units')
    self.use_bn = params.get_or_default('use_bn', True)
    self.use_final_bn = params.get_or_default('use_final_bn', True)
    self.use_bias = params.get_or_default('use_bias', False)
    self.use_final_bias = params.get_or_default('use_final_bias', False)
    self.dropout_rate = list(params.get_or_default('dropout_ratio', []))
    self.activation = params.get_or_default('activation', 'relu')
    self.initializer = params.get_or_default('initializer', 'he_uniform')
    self.final_activation = params.get_or_default('final_activation', None)
    assert not params.get_or_default('use_bn_after_activation', False), \
        'use_bn_after_activation is outside the hot-path scope'
    self.units = [int(u) for u in params.hidden_units]  # (st_params hold doubles; keras Dense takes int(units))
    assert len(self.units) > 0, 'MLP(%s) takes at least one hidden units' % name
    self.l2_reg = params.l2_regularizer
    self.add_to_outputs = params.get_or_default('add_to_outputs', False)
    logging.info('MLP(%s) units: %s, activate=%s, use_bn=%r, final_bn=%r, final_activate=%s, bias=%r, '
                 'initializer=%s' % (name, self.units, self.activation, self.use_bn, self.use_final_bn,
                                     self.final_activation, self.use_bias, self.initializer))

  def _layer(self, x, i, units, use_bn, act, use_bias, drop, training):
    lname = '%s/layer_%d' % (self.name, i)
    act = None if (act is None or str(act).lower() == 'linear') else act
    fuse_relu = act is not None and is_relu(act)
    x = dnn.dense_bn_act(x, units, lname + '/dense', self.l2_reg, use_bias, use_bn, fuse_relu, training,
                         bn_name=lname + '/bn', kernel_initializer=self.initializer)
    if act is not None and not fuse_relu:
      fn = get_activation(act, training=training) if str(act).lower() == 'dice' else get_activation(act)
      if fn is not None:
        x = fn(x, name=lname + '/act')
    if 0.0 < drop < 1.0 and training:
      x = torch.nn.functional.dropout(x, p=drop, training=True)
    elif drop >= 1.0:
      raise ValueError('invalid dropout_ratio: %.3f' % drop)
    return x

  def __call__(self, x, training=None, **kwargs):
    nd = len(self.dropout_rate)
    n = len(self.units) - 1
    for i, u in enumerate(self.units[:-1]):
      x = self._layer(x, i, u, self.use_bn, self.activation, self.use_bias,
                      self.dropout_rate[i] if i < nd else 0.0, training)
    x = self._layer(x, n, self.units[-1], self.use_final_bn, self.final_activation, self.use_final_bias,
                    self.dropout_rate[n] if nd > n else 0.0, training)
    if self.add_to_outputs and 'prediction_dict' in kwargs:
      kwargs['prediction_dict'][self.layer_name] = x.squeeze(1)
    return x


class Add(object):
  """tf.keras.layers.Add, which backbone configs name directly (`class_name: 'Add'` with `merge_inputs_into_list`,
  e.g. the final logit of examples/configs/wide_and_deep_backbone_on_movielens.config): the sum of a list of tensors."""

  def __init__(self, params=None, name=None, reuse=None, **kwargs):
    self.name = name

  def __call__(self, inputs, **kwargs):
    inputs = list(inputs)
    assert len(inputs) >= 2, 'A merge layer should be called on a list of at least 2 inputs'
    out = inputs[0]
    for t in inputs[1:]:
      out = out + t
    return out
