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


class DCN(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DCN, self).__init__(model_config, feature_configs, features, labels, is_training)
    self._take_config('dcn')

  def _cross_net(self, tensor, num_cross_layers):
    if num_cross_layers == 0:
      return tensor
    store, width = context.varstore(), tensor.shape[-1]
    # w and b of every layer, created in the reference's order (w_0, b_0, w_1, ...): both glorot-uniform (dcn.py:37-42)
    pairs = [(store.get_variable('cross_layer_%s_w' % i, (width,), 'glorot_uniform'),
              store.get_variable('cross_layer_%s_b' % i, (width,), 'glorot_uniform')) for i in range(num_cross_layers)]
    ws, bs = [w for w, _ in pairs], [b for _, b in pairs]
    if all(t.grad is not None for t in ws + bs):
      # the stacks are detached copies: gradients go straight into the variables' gradient slices
      return kernels.CrossV1Fn.apply(tensor, torch.stack([t.detach() for t in ws]), torch.stack([t.detach() for t in bs]),
                                     ([t.grad for t in ws], [t.grad for t in bs]))
    return kernels.CrossV1Fn.apply(tensor, torch.stack(ws), torch.stack(bs))

  def build_predict_graph(self):
    own = self._model_config
    self._features = self._group('all')[0]
    deep = self._dnn(self._features, own.deep_tower.dnn, 'dnn')
    crossed = self._cross_net(self._features, own.cross_tower.cross_num)
    top = self._dnn(torch.cat([deep, crossed], dim=1), own.final_dnn, 'final_dnn')
    return self._emit(dnn.dense(top, self._num_class, 'output', head=True))  # (no kernel regulariser on `output`: dcn.py:66)
