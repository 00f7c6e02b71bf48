"""DLRM (reference easy_rec/python/model/dlrm.py:15-73).

bot_dnn over the `dense` group; arch_interaction_op 'cat': concat[bot, sparse embeddings]; 'dot': every pairwise dot
product of [bot] + sparse embeddings (upper triangle, with the diagonal if arch_interaction_itself), followed by the
sparse embeddings and - with arch_with_dense_feature - the bot output; top_dnn; dense(1).  The einsum + F slices +
concat of the reference are ONE launch each way (er_dot_interaction_fwd / _bwd).
"""
import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos.dlrm_pb2 import DLRM as DLRMConfig


class DLRM(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DLRM, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert model_config.WhichOneof('model') == 'dlrm', 'invalid model config: %s' % model_config.WhichOneof('model')
    self._model_config = model_config.dlrm
    assert isinstance(self._model_config, DLRMConfig)
    assert self._input_layer.has_group('sparse'), 'sparse group is not specified'
    assert self._input_layer.has_group('dense'), 'dense group is not specified'

  def build_predict_graph(self):
    sparse_cat, sparse_features = self._input_layer(self._feature_dict, 'sparse')
    dense_feature, _ = self._input_layer(self._feature_dict, 'dense')
    bot_dnn = dnn.DNN(self._model_config.bot_dnn, self._l2_reg, 'bot_dnn', self._is_training)
    dense_fea = bot_dnn(dense_feature)
    op = self._model_config.arch_interaction_op
    if op == 'cat':
      all_fea = torch.cat([dense_fea, sparse_cat], dim=1)
    elif op == 'dot':
      E = sparse_features[0].shape[1]
      assert dense_fea.shape[1] == E, 'bot_dnn last hidden[%d] != sparse feature embedding_dim[%d]' % (
          dense_fea.shape[1], E)
      assert all(f.shape[1] == E for f in sparse_features)
      all_feas = torch.cat([dense_fea, sparse_cat], dim=1)  # [B, (1 + n_sparse) * E]
      num_fea = 1 + len(sparse_features)
      upper_tri = kernels.DotInteractionFn.apply(all_feas, num_fea, E,
                                                 bool(self._model_config.arch_interaction_itself))
      concat_feas = [upper_tri, sparse_cat]
      if self._model_config.arch_with_dense_feature:
        concat_feas.append(dense_fea)
      all_fea = torch.cat(concat_feas, dim=1)
    else:
      raise ValueError('invalid arch_interaction_op: %s' % op)
    top_dnn = dnn.DNN(self._model_config.top_dnn, self._l2_reg, 'top_dnn', self._is_training)
    all_fea = top_dnn(all_fea)
    logits = dnn.dense(all_fea, 1, 'output', l2_reg=self._l2_reg)
    self._add_to_prediction_dict(logits)
    return self._prediction_dict
