"""Target attention (DIN) over the sequence features declared INSIDE a feature group.

API of reference easy_rec/python/layers/sequence_feature_layer.py:16-270: `SequenceFeatureLayer(feature_configs,
feature_groups_config, ...)` collects every group's `sequence_features` (SeqAttGroupConfig entries); called with the
group's concatenated plain features it returns (concat_features, [one attention output per entry]).  Keys may be
the embeddings the enclosing group already produced (`feature_name_to_output_tensors`), the history tables live under
the enclosing group's name scope; the attention MLP is `seq_dnn` (default hidden units 128-64-32-1) named
`seq_dnn<group_name>`, last layer without BatchNorm / activation (:165-172).

`target_attention` is the same math as MultiTowerDIN.din (model/multi_tower_din.py:62-97), with two fused HIP kernels
around the attention MLP: `er_din_concat` builds [q, h, q - h, q * h], `er_din_pool` masks, soft-maxes and pools.
Negative-sampler attention is outside the hot-path scope.
"""
import logging

import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.layers import seq_input_layer


def target_attention(dnn_config, deep_fea, name, l2_reg, is_training, need_key_feature=True, allow_key_transform=False,
                     transform_dnn=False):
  """deep_fea: SeqInputLayer's dict.  Returns [B, E_hist (+ E_key)]."""
  cur_id, hist, seq_len = deep_fea['key'], deep_fea['hist_seq_emb'], deep_fea['hist_seq_len']
  assert not deep_fea['aux_hist_seq_emb_list'], 'aux_hist_seq is outside the hot-path scope'
  B, L, E = hist.shape
  if allow_key_transform and cur_id.shape[-1] != E:
    if E > cur_id.shape[-1] and not transform_dnn:  # zero-pad the key up to the history's width (:139-140)
      cur_id = torch.nn.functional.pad(cur_id, (0, E - cur_id.shape[-1]))
    else:
      cur_id = dnn.dense(cur_id, E, 'sequence_key_transform_layer_' + name)
      hist = dnn.dense(hist, E, 'sequence_fea_transform_layer_' + name)
  assert cur_id.shape[1] == E, 'DIN: key dim %d != history dim %d (set allow_key_transform)' % (cur_id.shape[1], E)
  hist = hist if hist.is_contiguous() else hist.contiguous()  # (a batch whose longest sequence is below max_seq_len)
  hist = kernels.slot_gate(hist)  # (DINConcatFn and DINPoolFn share the history's gradient buffer)
  din_layer = dnn.DNN(dnn_config, l2_reg, name, is_training, last_layer_no_activation=True,
                      last_layer_no_batch_norm=True)
  if dnn.din_first_layer_ok(din_layer, cur_id, hist):
    # the [B, L, 4E] attention input is never built: the first layer's contractions generate [q, h, q - h, q * h] from
    # (q, h) while staging and reduce its gradient to dq / dh in their epilogue (kernels.DINFirstLayerFn, round 6; the
    # ALGEBRAIC fold of round 4 - a different thing - was measured slower and removed: profiles/r04_din_folded_first_layer_ab.txt)
    scores = din_layer(None, din=(cur_id, hist)).reshape(B, L)
  else:
    din_net = kernels.DINConcatFn.apply(cur_id, hist)  # [B, L, 4E]
    scores = din_layer(din_net).reshape(B, L)
  pooled = kernels.DINPoolFn.apply(scores, hist, seq_len, 1.0)  # softmax over where(t < len, score, -2^32 + 1)
  if not need_key_feature:
    return pooled
  return kernels.concat_cols([pooled, cur_id])


class SequenceFeatureLayer(object):

  def __init__(self, feature_configs, feature_groups_config, ev_params=None, embedding_regularizer=None,
               kernel_regularizer=None, is_training=False, is_predicting=False, engine=None):
    self._seq_feature_groups_config = [y for x in feature_groups_config for y in x.sequence_features]
    self._seq_input_layer = None
    if self._seq_feature_groups_config:
      self._seq_input_layer = seq_input_layer.SeqInputLayer(
          feature_configs, self._seq_feature_groups_config, embedding_regularizer=embedding_regularizer,
          ev_params=ev_params, engine=engine)
    self._embedding_regularizer = embedding_regularizer
    self._kernel_regularizer = kernel_regularizer
    self._is_training = is_training
    self._is_predicting = is_predicting

  def target_attention(self, dnn_config, deep_fea, name, need_key_feature=True, allow_key_transform=False,
                       transform_dnn=False):
    return target_attention(dnn_config, deep_fea, name, self._kernel_regularizer, self._is_training,
                            need_key_feature, allow_key_transform, transform_dnn)

  def __call__(self, features, concat_features, all_seq_att_map_config, feature_name_to_output_tensors=None,
               negative_sampler=False, scope_name=None):
    assert not negative_sampler, 'negative-sampler target attention is outside the hot-path scope'
    logging.info('use sequence feature layer.')
    all_seq_fea = []
    for cfg in all_seq_att_map_config:
      seq_features = self._seq_input_layer(features, cfg.group_name, feature_name_to_output_tensors,
                                           cfg.allow_key_search, scope_name, requires_grad=self._is_training)
      if cfg.HasField('seq_dnn'):
        seq_dnn_config = cfg.seq_dnn
      else:
        logging.info('seq_dnn not set in seq_att_groups, will use default settings')
        from easyrec_amd.protos.dnn_pb2 import DNN
        seq_dnn_config = DNN()
        seq_dnn_config.hidden_units.extend([128, 64, 32, 1])
      all_seq_fea.append(self.target_attention(
          seq_dnn_config, seq_features, name='seq_dnn' + cfg.group_name, need_key_feature=cfg.need_key_feature,
          allow_key_transform=cfg.allow_key_transform, transform_dnn=cfg.transform_dnn))
    return concat_features, all_seq_fea
