"""WideAndDeep (reference easy_rec/python/model/wide_and_deep.py:17-86).

wide = add_n of the wide group's per-feature embeddings (dimension wide_output_dim); deep = DNN(deep concat);
with a final_dnn: dense(num_class) over final_dnn(concat[wide, deep]); without: dense(deep) + wide, the wide
embeddings then having num_class columns (:38-45).  Same kernels as DeepFM: the two groups read the same id columns
and share one sort per step (er_emb_group_share_sort), the wide sum is er_rowsum_fwd.
"""
import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos.wide_and_deep_pb2 import WideAndDeep as WideAndDeepConfig


class WideAndDeep(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(WideAndDeep, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert model_config.WhichOneof('model') == 'wide_and_deep', \
        'invalid model config: %s' % model_config.WhichOneof('model')
    self._model_config = model_config.wide_and_deep
    assert isinstance(self._model_config, WideAndDeepConfig)
    assert self._input_layer.has_group('wide')
    assert self._input_layer.has_group('deep')

  def build_input_layer(self, model_config, feature_configs):
    # overwrite create input_layer to support wide_output_dim (wide_and_deep.py:38-45)
    has_final = len(model_config.wide_and_deep.final_dnn.hidden_units) > 0
    self._wide_output_dim = model_config.wide_and_deep.wide_output_dim
    if not has_final:
      model_config.wide_and_deep.wide_output_dim = model_config.num_class
      self._wide_output_dim = model_config.num_class
    super(WideAndDeep, self).build_input_layer(model_config, feature_configs)

  def build_predict_graph(self):
    wide_cat, wide_features = self._input_layer(self._feature_dict, 'wide')
    deep_features, _ = self._input_layer(self._feature_dict, 'deep')
    wd = self._wide_output_dim
    if wd == 1:
      wide_fea = kernels.RowSumFn.apply(wide_cat, kernels.grad_sink_of(wide_cat))  # add_n of [B, 1] columns
    else:
      wide_fea = wide_cat.reshape(wide_cat.shape[0], len(wide_features), wd).sum(dim=1)

    deep_layer = dnn.DNN(self._model_config.dnn, self._l2_reg, 'deep_feature', self._is_training)
    deep_fea = deep_layer(deep_features)

    if len(self._model_config.final_dnn.hidden_units) > 0:
      all_fea = kernels.concat_cols([wide_fea, deep_fea])
      final_layer = dnn.DNN(self._model_config.final_dnn, self._l2_reg, 'final_dnn', self._is_training)
      all_fea = final_layer(all_fea)
      output = dnn.dense(all_fea, self._num_class, 'output', l2_reg=self._l2_reg)
    else:
      deep_out = dnn.dense(deep_fea, self._num_class, 'deep_out', l2_reg=self._l2_reg)
      output = deep_out + wide_fea

    self._add_to_prediction_dict(output)
    return self._prediction_dict
