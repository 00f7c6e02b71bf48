"""MultiTowerDIN (reference easy_rec/python/model/multi_tower_din.py:18-130).

Plain towers: BatchNorm on the tower input (`<tower>_fea_bn`) -> DNN.  DIN towers: target attention
  a[b,t] = MLP([q_b, h_bt, q_b - h_bt, q_b * h_bt])   (BN in the MLP sees the padded positions too, :77-85)
  p = softmax over t of where(t < len_b, a, -2^32+1); out = concat[p @ h, q]                 (:86-96)
with two fused HIP kernels around the attention MLP instead of Tile/Concat/SequenceMask/Select/Softmax/
BatchMatMul: `er_din_concat` builds the MLP input, `er_din_pool` masks, soft-maxes and pools.
"""
import logging

import torch

from easyrec_amd import kernels

from easyrec_amd.core import context
from easyrec_amd.layers import dnn
from easyrec_amd.layers import seq_input_layer
from easyrec_amd.layers.sequence_feature_layer import target_attention
from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos.multi_tower_pb2 import MultiTower as MultiTowerConfig


class MultiTowerDIN(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(MultiTowerDIN, self).__init__(model_config, feature_configs, features, labels, is_training)
    self._seq_input_layer = seq_input_layer.SeqInputLayer(
        feature_configs, model_config.seq_att_groups, embedding_regularizer=self._emb_reg,
        ev_params=self._global_ev_params, engine=context.current().engine)
    assert self._model_config.WhichOneof('model') == 'multi_tower', \
        'invalid model config: %s' % self._model_config.WhichOneof('model')
    self._model_config = self._model_config.multi_tower
    assert isinstance(self._model_config, MultiTowerConfig)
    assert len(self._model_config.bst_towers) == 0, 'bst_towers are outside the hot-path scope'
    self._tower_num = len(self._model_config.towers)
    self._din_tower_num = len(self._model_config.din_towers)
    logging.info('all tower num: {0}'.format(self._tower_num + self._din_tower_num))
    logging.info('din tower num: {0}'.format(self._din_tower_num))

  def din(self, dnn_config, deep_fea, name):
    # [q, h, q - h, q * h] -> attention MLP (BatchNorm over B x L positions, L = the batch's longest sequence) ->
    # masked softmax -> pooled history, concatenated with the key
    return target_attention(dnn_config, deep_fea, name, self._l2_reg, self._is_training)

  def build_predict_graph(self):
    # input layer calls in the reference's constructor order: plain towers, then DIN towers
    tower_features = []
    for tower in self._model_config.towers:
      tower_feature, _ = self._input_layer(self._feature_dict, tower.input)
      tower_features.append(tower_feature)
    din_features = []
    for tower in self._model_config.din_towers:
      din_features.append(self._seq_input_layer(self._feature_dict, tower.input, requires_grad=self._is_training))

    # the plain towers are independent stacks: layer by layer in grouped launches when their depths agree
    # (dnn.run_parallel; the reference's order - one tower after the other - otherwise)
    stacks, inputs = [], []
    for tower, tower_fea in zip(self._model_config.towers, tower_features):
      tower_name = tower.input
      inputs.append(dnn.batch_norm(tower_fea, '%s_fea_bn' % tower_name, self._is_training))
      stacks.append(dnn.DNN(tower.dnn, self._l2_reg, '%s_dnn' % tower_name, self._is_training))
    tower_fea_arr, _ = dnn.run_parallel(stacks, inputs) if stacks else ([], [])
    for tower, tower_fea in zip(self._model_config.din_towers, din_features):
      tower_fea_arr.append(self.din(tower.dnn, tower_fea, name='%s_dnn' % tower.input))

    all_fea = kernels.concat_cols(tower_fea_arr)
    final_dnn_layer = dnn.DNN(self._model_config.final_dnn, self._l2_reg, 'final_dnn', self._is_training)
    all_fea = final_dnn_layer(all_fea)
    output = dnn.dense(all_fea, self._num_class, 'output', head=True)
    self._add_to_prediction_dict(output)
    return self._prediction_dict
