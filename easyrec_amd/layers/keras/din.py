"""Keras-style `DIN` block of the backbone API (reference easy_rec/python/layers/keras/din.py:13-67).

Input: (history [B, L, E], lengths [B], target [B, E']) - an `input_layer { output_seq_and_normal_feature: true }`
block.  Attention scores = MLP `din_attention` over [q, h, q - h, q * h] with the last layer forced to bias / no
BatchNorm / linear (:19-22); padded positions get -2^32 + 1; `attention_normalizer` softmax (default) or sigmoid of
score / sqrt(E); output = scores @ history (+ the target when `need_target_feature`).  A target narrower than the
history is zero-padded for the attention (and appended in that padded form) and the pooled history is cut back to the
target's width (:31-40, 60-64).

Kernels: `er_din_concat` builds the MLP input in one pass, `er_din_pool` does mask + softmax + pooling (forward and
backward); the sigmoid normaliser (no shipped hot-path config uses it) is plain elementwise torch.
"""
import logging

import torch

from easyrec_amd import kernels
from easyrec_amd.layers.keras.blocks import MLP
from easyrec_amd.layers.utils import Parameter

MASK_VALUE = float(-2**32 + 1)


class DIN(object):

  def __init__(self, params, name='din', reuse=None, **kwargs):
    self.name = name
    self.l2_reg = params.l2_regularizer
    self.config = params.get_pb_config()
    att = self.config.attention_dnn
    att.use_final_bn, att.use_final_bias, att.final_activation = False, True, 'linear'
    mlp_params = Parameter.make_from_pb(att)
    mlp_params.l2_regularizer = self.l2_reg
    # a layer called inside another layer's call() creates its variables under BOTH name scopes (keras): two DIN
    # blocks of one backbone keep separate attention MLPs, `<block>/din_attention/layer_i/...`
    self.din_layer = MLP(mlp_params, name + '/din_attention', reuse=reuse)

  def __call__(self, inputs, training=None, **kwargs):
    keys, seq_len, query = inputs
    assert query is not None, '[%s] target feature is empty' % self.name
    B, L, E = keys.shape
    q_dim = int(query.shape[-1])
    if q_dim != E:
      logging.info('<din> the embedding size of sequence [%d] and target item [%d] is not equal in feature group: %s',
                   E, q_dim, self.name)
      assert q_dim < E, 'the embedding size of target item is larger than the one of sequence'
    q_att = query if q_dim == E else torch.nn.functional.pad(query, (0, E - q_dim))
    keys = keys if keys.is_contiguous() else keys.contiguous()  # (a batch shorter than max_seq_len is a slice)
    keys = kernels.slot_gate(keys)  # (DINConcatFn and DINPoolFn share the history's gradient buffer)
    din_all = kernels.DINConcatFn.apply(q_att, keys)  # [B, L, 4E]
    scores = self.din_layer(din_all, training=training).reshape(B, L)
    norm = self.config.attention_normalizer
    if norm == 'softmax':
      pooled = kernels.DINPoolFn.apply(scores, keys, seq_len, 1.0)  # [B, E]
    elif norm == 'sigmoid':
      mask = torch.arange(L, device=keys.device).view(1, L) < seq_len.view(B, 1)
      w = torch.sigmoid(torch.where(mask, scores, torch.full_like(scores, MASK_VALUE)) / (E ** 0.5))
      pooled = torch.bmm(w.view(B, 1, L), keys).view(B, E)
    else:
      raise ValueError('unsupported attention normalizer: ' + norm)
    if q_dim < E:
      pooled = pooled[:, :q_dim]
    if self.config.need_target_feature:
      # (the reference re-binds `query` to its zero-padded form at :36, so a narrower target is appended PADDED to the
      # history's width - pinned by tests/test_reference_layers.py against the reference's own DIN.call)
      pooled = torch.cat([pooled, q_att], dim=-1)
    return pooled
