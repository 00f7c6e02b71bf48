"""WideAndDeep (reference easy_rec/python/model/wide_and_deep.py:17-86).

wide = add_n of the wide group's per-feature embeddings (dimension wide_output_dim); deep = DNN(deep concat);
with a final_dnn: dense(num_class) over final_dnn(concat[wide, deep]); without: dense(deep) + wide, the wide
embeddings then having num_class columns (:38-45).  Same kernels as DeepFM: the two groups read the same id columns
and share one sort per step (er_emb_group_share_sort), the wide sum is er_rowsum_fwd.
"""
from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.model.rank_model import RankModel


class WideAndDeep(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(WideAndDeep, self).__init__(model_config, feature_configs, features, labels, is_training)
    self._take_config('wide_and_deep')
    for needed in ('wide', 'deep'):
      assert self._input_layer.has_group(needed), needed + ' group is not specified'

  def build_input_layer(self, model_config, feature_configs):
    own = model_config.wide_and_deep
    if len(own.final_dnn.hidden_units) == 0:
      own.wide_output_dim = model_config.num_class  # (the wide sum IS a logit then: wide_and_deep.py:38-45)
    self._wide_output_dim = own.wide_output_dim
    super(WideAndDeep, self).build_input_layer(model_config, feature_configs)

  def build_predict_graph(self):
    wide_block, wide_list = self._group('wide')
    deep_block = self._group('deep')[0]
    width = self._wide_output_dim
    if width == 1:  # add_n of [B, 1] columns = a row sum of the group's block
      wide = kernels.RowSumFn.apply(wide_block, kernels.grad_sink_of(wide_block))
    else:
      wide = wide_block.reshape(wide_block.shape[0], len(wide_list), width).sum(dim=1)
    deep = self._dnn(deep_block, self._model_config.dnn, 'deep_feature')
    if len(self._model_config.final_dnn.hidden_units) > 0:
      top = self._dnn(kernels.concat_cols([wide, deep]), self._model_config.final_dnn, 'final_dnn')
      return self._emit(dnn.dense(top, self._num_class, 'output', l2_reg=self._l2_reg, head=True))
    return self._emit(dnn.dense(deep, self._num_class, 'deep_out', l2_reg=self._l2_reg) + wide)
