"""FM (reference easy_rec/python/model/fm.py:15-63): wide sum + second-order FM over the deep group + a bias.

num_class == 1: logits = sum(wide) + sum_d FM(deep)[d] + fm_bias; otherwise the FM vector goes through
dense(num_class, name='fm_logits').
"""
from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.layers import dnn
from easyrec_amd.layers import fm
from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos.fm_pb2 import FM as FMConfig


class FM(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(FM, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert self._model_config.WhichOneof('model') == 'fm', \
        'invalid model config: %s' % self._model_config.WhichOneof('model')
    self._model_config = self._model_config.fm
    assert isinstance(self._model_config, FMConfig)

  def build_input_layer(self, model_config, feature_configs):
    # overwrite create input_layer to support wide_output_dim (fm.py:35-38)
    self._wide_output_dim = model_config.num_class
    super(FM, self).build_input_layer(model_config, feature_configs)

  def build_predict_graph(self):
    wide_features, _ = self._input_layer(self._feature_dict, 'wide')
    _, fm_features = self._input_layer(self._feature_dict, 'deep')
    assert self._num_class == 1 or self._wide_output_dim == 1, 'multi-class wide sum: outside the hot-path scope'
    wide_fea = kernels.RowSumFn.apply(wide_features, kernels.grad_sink_of(wide_features))
    fm_fea = fm.FM(name='fm_feature')(fm_features)
    if self._num_class > 1:
      fm_fea = dnn.dense(fm_fea, self._num_class, 'fm_logits', l2_reg=self._l2_reg)
    else:
      fm_fea = kernels.RowSumFn.apply(fm_fea)
    bias = context.varstore().get_variable('fm_bias', (self._num_class,), 'zeros')
    output = (wide_fea + fm_fea) + bias
    self._add_to_prediction_dict(output)
    return self._prediction_dict
