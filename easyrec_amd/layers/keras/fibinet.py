"""`SENet` of FiBiNET as a backbone block (reference easy_rec/python/layers/keras/fibinet.py:15-92): a list of field
embeddings -> per field and squeeze group the max and the mean over the group's columns -> Dense(relu) ->
Dense(sum of dims) -> re-weight the concatenated embeddings (+ skip connection, + LayerNormalization).

Not a hot-path layer (SURVEY.md section 2 row 8 lists the FiBiNET family as out of scope); it is here because the
reference's own fixture `samples/model_config/mmoe_backbone_on_taobao.config` puts it in front of MMoE.  The two
dense layers are the library's MFMA GEMMs; squeeze / re-weight / layer-norm are elementwise torch ops.
"""
import math

import torch

from easyrec_amd.core import context
from easyrec_amd.core.variables import truncated_normal
from easyrec_amd.layers import dnn

LN_EPSILON = 1e-3  # keras LayerNormalization default
_TRUNC_STD = 0.87962566103423978  # std of a unit normal truncated to 2 sigma (keras VarianceScaling divides by it)


def _he_normal(shape, rng):
  return truncated_normal(shape, rng, 0.0, math.sqrt(2.0 / shape[0]) / _TRUNC_STD)


def _glorot_normal(shape, rng):
  return truncated_normal(shape, rng, 0.0, math.sqrt(2.0 / (shape[0] + shape[1])) / _TRUNC_STD)


class SENet(object):

  def __init__(self, params, name='SENet', reuse=None, **kwargs):
    self.name = name
    self.config = params.get_pb_config()

  def __call__(self, inputs, **kwargs):
    g = int(self.config.num_squeeze_group)
    inputs = list(inputs)
    for emb in inputs:
      assert emb.dim() == 2, 'field embeddings must be rank 2 tensors'
      d = int(emb.shape[-1])
      assert d >= g and d % g == 0, 'field embedding dimension %d must be divisible by %d' % (d, g)
    emb_size = sum(int(e.shape[-1]) for e in inputs)
    squeezed = []
    for emb in inputs:
      grouped = emb.reshape(emb.shape[0], g, -1)
      squeezed.append(grouped.max(dim=-1).values)
      squeezed.append(grouped.mean(dim=-1))
    z = torch.cat(squeezed, dim=1)  # [B, fields * groups * 2]
    reduction = max(1, len(inputs) * g * 2 // int(self.config.reduction_ratio))
    a1 = torch.relu(dnn.dense(z, reduction, self.name + '/W1', kernel_initializer=_he_normal))
    weights = dnn.dense(a1, emb_size, self.name + '/W2', kernel_initializer=_glorot_normal)
    x = torch.cat(inputs, dim=-1)
    out = x * weights
    if self.config.use_skip_connection:
      out = out + x
    if self.config.use_output_layer_norm:
      vs = context.varstore()
      gamma = vs.get_variable(self.name + '/output_ln/gamma', (emb_size,), 'ones')
      beta = vs.get_variable(self.name + '/output_ln/beta', (emb_size,), 'zeros')
      mean = out.mean(dim=-1, keepdim=True)
      var = ((out - mean) ** 2).mean(dim=-1, keepdim=True)
      out = (out - mean) * torch.rsqrt(var + LN_EPSILON) * gamma + beta
    return out
