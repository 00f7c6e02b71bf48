"""DeepFM (reference easy_rec/python/model/deepfm.py:17-109).

wide = sum over the wide group's 1-dim embeddings; fm = FM over the deep (or `fm`) group's field
embeddings kept as [B, D]; deep = DNN(deep concat); final_dnn over concat[wide, fm, deep] -> dense(1).
"""
import logging

import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.layers import fm
from easyrec_amd.model.rank_model import RankModel


class DeepFM(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DeepFM, self).__init__(model_config, feature_configs, features, labels, is_training)
    if self._take_config('deepfm').HasField('wide_regularization'):
      logging.warning('wide_regularization is deprecated, please use l2_regularization')

  def build_input_layer(self, model_config, feature_configs):
    # the wide columns' width comes from the model's own config (deepfm.py:42-51)
    self._wide_output_dim = model_config.deepfm.wide_output_dim
    if len(model_config.deepfm.final_dnn.hidden_units) == 0:
      assert self._wide_output_dim == model_config.num_class
    elif self._wide_output_dim != model_config.num_class:
      logging.warning('wide_output_dim not equal to 1, it is not a standard model')
    super(DeepFM, self).build_input_layer(model_config, feature_configs)

  def build_predict_graph(self):
    # the input-layer calls in the reference's order - wide, deep[, fm] (deepfm.py:36-40)
    self._wide_features = self._group('wide')[0]
    self._deep_features, self._fm_features = self._group('deep')
    if self._input_layer.has_group('fm'):
      self._fm_features = self._group('fm')[1]
    if self._num_class > 1 and self._wide_output_dim == self._num_class:
      raise AssertionError('multi-class wide output is outside the hot-path scope')
    own = self._model_config
    # [reduce_sum(wide) | FM | deep] in one launch when both blocks are embedding group outputs of a training step
    # (kernels.WideFmConcatFn); else the three separate ops
    wide_sink = kernels.grad_sink_of(self._wide_features)
    blk = fm.FM.group_block(self._fm_features)
    fused = (len(own.final_dnn.hidden_units) > 0 and getattr(kernels.hip(), 'fused_wide_fm', False) and self._is_training and
             torch.is_grad_enabled() and wide_sink is not None and blk is not None and blk[3] is not None and
             self._wide_features.dim() == 2 and self._wide_features.stride(-1) == 1)
    if fused:
      # (the tower's last BatchNorm finalize + apply is left to the concat launch: kernels.LinearBNActFn(defer_apply))
      deep = self._dnn(self._deep_features, own.dnn, 'deep_feature', defer_last_apply=True)
      x, F, D, fm_sink, col0 = blk
      joined = kernels.WideFmConcatFn.apply(self._wide_features, x, deep, F, D, wide_sink, fm_sink, col0)
      kernels.finish_pending_bn(deep)  # (a no-op: WideFmConcatFn ran it)
      kernels.tag_bn_cols(joined, deep, 1 + D)  # (the deep tower's last BatchNorm backward: sums from final_dnn's dgrad)
      self._fm_outputs = joined[:, 1:1 + D]
      top = self._dnn(joined, own.final_dnn, 'final_dnn')
      return self._emit(dnn.dense(top, self._num_class, 'output', l2_reg=self._l2_reg, head=True))
    wide = kernels.RowSumFn.apply(self._wide_features, wide_sink)
    self._fm_outputs = pairwise = fm.FM(name='fm_feature')(self._fm_features)
    deep = self._dnn(self._deep_features, own.dnn, 'deep_feature')
    if len(own.final_dnn.hidden_units) > 0:
      top = self._dnn(kernels.concat_cols([wide, pairwise, deep]), own.final_dnn, 'final_dnn')
      return self._emit(dnn.dense(top, self._num_class, 'output', l2_reg=self._l2_reg, head=True))
    # without a final_dnn the three parts ARE logits and add up (deepfm.py:90-105)
    deep_logit = dnn.dense(deep, self._num_class, 'deep_logits', l2_reg=self._l2_reg)
    return self._emit(wide + pairwise.sum(dim=1, keepdim=True) + deep_logit)
