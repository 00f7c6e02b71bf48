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
from easyrec_amd.protos.deepfm_pb2 import DeepFM as DeepFMConfig


class DeepFM(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DeepFM, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert self._model_config.WhichOneof('model') == 'deepfm', \
        'invalid model config: %s' % self._model_config.WhichOneof('model')
    self._model_config = self._model_config.deepfm
    assert isinstance(self._model_config, DeepFMConfig)
    if self._model_config.HasField('wide_regularization'):
      logging.warning('wide_regularization is deprecated, please use l2_regularization')

  def build_input_layer(self, model_config, feature_configs):
    # overwrite create input_layer to support wide_output_dim (reference deepfm.py:42-51)
    self._wide_output_dim = model_config.deepfm.wide_output_dim
    has_final = len(model_config.deepfm.final_dnn.hidden_units) > 0
    if not has_final:
      assert self._wide_output_dim == model_config.num_class
    elif self._wide_output_dim != model_config.num_class:
      logging.warning('wide_output_dim not equal to 1, it is not a standard model')
    super(DeepFM, self).build_input_layer(model_config, feature_configs)

  def build_predict_graph(self):
    # input layer calls in the reference's order (wide, deep[, fm]): deepfm.py:36-40
    self._wide_features, _ = self._input_layer(self._feature_dict, 'wide')
    self._deep_features, self._fm_features = self._input_layer(self._feature_dict, 'deep')
    if self._input_layer.has_group('fm'):
      _, self._fm_features = self._input_layer(self._feature_dict, 'fm')

    # Wide
    assert not (self._num_class > 1 and self._wide_output_dim == self._num_class), \
        'multi-class wide output is outside the hot-path scope'
    wide_fea = kernels.RowSumFn.apply(self._wide_features, kernels.grad_sink_of(self._wide_features))

    # FM
    fm_fea = fm.FM(name='fm_feature')(self._fm_features)
    self._fm_outputs = fm_fea

    # Deep
    deep_layer = dnn.DNN(self._model_config.dnn, self._l2_reg, 'deep_feature', self._is_training)
    deep_fea = deep_layer(self._deep_features)

    # Final
    if len(self._model_config.final_dnn.hidden_units) > 0:
      all_fea = kernels.concat_cols([wide_fea, fm_fea, deep_fea])
      final_dnn_layer = dnn.DNN(self._model_config.final_dnn, self._l2_reg, 'final_dnn', self._is_training)
      all_fea = kernels.mark_single_consumer(final_dnn_layer(all_fea))  # read by the `output` projection alone
      output = dnn.dense(all_fea, self._num_class, 'output', l2_reg=self._l2_reg)
    else:
      fm_sum = fm_fea.sum(dim=1, keepdim=True)
      deep_logit = dnn.dense(deep_fea, self._num_class, 'deep_logits', l2_reg=self._l2_reg)
      output = wide_fea + fm_sum + deep_logit

    self._add_to_prediction_dict(output)
    return self._prediction_dict
