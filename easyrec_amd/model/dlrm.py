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


class DLRM(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DLRM, self).__init__(model_config, feature_configs, features, labels, is_training)
    self._take_config('dlrm')
    for needed in ('sparse', 'dense'):
      assert self._input_layer.has_group(needed), needed + ' group is not specified'

  def _interact(self, bottom, sparse_block, sparse_list):
    own = self._model_config
    if own.arch_interaction_op == 'cat':
      return torch.cat([bottom, sparse_block], dim=1)
    if own.arch_interaction_op != 'dot':
      raise ValueError('invalid arch_interaction_op: %s' % own.arch_interaction_op)
    width = sparse_list[0].shape[1]
    assert bottom.shape[1] == width, 'bot_dnn last hidden[%d] != sparse feature embedding_dim[%d]' % (bottom.shape[1], width)
    assert all(f.shape[1] == width for f in sparse_list)
    fields = torch.cat([bottom, sparse_block], dim=1)  # [B, (1 + n_sparse) * width], the bottom output first
    pairs = kernels.DotInteractionFn.apply(fields, 1 + len(sparse_list), width, bool(own.arch_interaction_itself))
    parts = [pairs, sparse_block] + ([bottom] if own.arch_with_dense_feature else [])
    return torch.cat(parts, dim=1)

  def build_predict_graph(self):
    sparse_block, sparse_list = self._group('sparse')
    dense_block = self._group('dense')[0]
    bottom = self._dnn(dense_block, self._model_config.bot_dnn, 'bot_dnn')
    top = self._dnn(self._interact(bottom, sparse_block, sparse_list), self._model_config.top_dnn, 'top_dnn')
    return self._emit(dnn.dense(top, 1, 'output', l2_reg=self._l2_reg, head=True))
