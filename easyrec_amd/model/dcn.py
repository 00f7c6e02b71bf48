"""DCN v1 (reference easy_rec/python/model/dcn.py:15-70).

deep tower = DNN over the `all` group; cross tower = `cross_num` layers
x_{l+1} = x0 * (x_l . w_l) + b_l + x_l  with w_l, b_l in R^d (tf.get_variable default initializer =
glorot_uniform for BOTH, dcn.py:37-42; no kernel regulariser on them nor on the `output` layer, :66);
final_dnn over concat[deep, cross] -> dense(num_class).  All cross layers run in ONE launch
(`er_cross_v1_fwd`, one wave per row) instead of 3 elementwise ops per layer.
"""
import torch

from easyrec_amd import kernels
from easyrec_amd.core import context
from easyrec_amd.layers import dnn
from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos.dcn_pb2 import DCN as DCNConfig


class DCN(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DCN, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert self._model_config.WhichOneof('model') == 'dcn', \
        'invalid model config: %s' % self._model_config.WhichOneof('model')
    self._model_config = self._model_config.dcn
    assert isinstance(self._model_config, DCNConfig)

  def _cross_net(self, tensor, num_cross_layers):
    vs = context.varstore()
    d = tensor.shape[-1]
    ws, bs = [], []
    for i in range(num_cross_layers):
      name = 'cross_layer_%s' % i
      ws.append(vs.get_variable(name + '_w', (d,), 'glorot_uniform'))
      bs.append(vs.get_variable(name + '_b', (d,), 'glorot_uniform'))
    if num_cross_layers == 0:
      return tensor
    bufs = None
    if all(t.grad is not None for t in ws + bs):
      bufs = ([t.grad for t in ws], [t.grad for t in bs])
      # the stacks are detached copies: gradients go straight into the variables' gradient slices
      return kernels.CrossV1Fn.apply(tensor, torch.stack([t.detach() for t in ws]),
                                     torch.stack([t.detach() for t in bs]), bufs)
    return kernels.CrossV1Fn.apply(tensor, torch.stack(ws), torch.stack(bs))

  def build_predict_graph(self):
    self._features, _ = self._input_layer(self._feature_dict, 'all')
    tower_fea_arr = []
    deep_tower_config = self._model_config.deep_tower
    dnn_layer = dnn.DNN(deep_tower_config.dnn, self._l2_reg, 'dnn', self._is_training)
    tower_fea_arr.append(dnn_layer(self._features))
    cross_tensor = self._cross_net(self._features, self._model_config.cross_tower.cross_num)
    tower_fea_arr.append(cross_tensor)
    all_fea = torch.cat(tower_fea_arr, dim=1)
    final_dnn_layer = dnn.DNN(self._model_config.final_dnn, self._l2_reg, 'final_dnn', self._is_training)
    all_fea = final_dnn_layer(all_fea)
    output = dnn.dense(all_fea, self._num_class, 'output')
    self._add_to_prediction_dict(output)
    return self._prediction_dict
