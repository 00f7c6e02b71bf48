"""MultiTower (reference easy_rec/python/model/multi_tower.py:15-62): per feature group a BatchNorm + DNN tower,
final_dnn over the concatenated tower outputs, dense(num_class) WITHOUT kernel regulariser (:58)."""
import torch

from easyrec_amd.layers import dnn
from easyrec_amd.model.rank_model import RankModel


class MultiTower(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(MultiTower, self).__init__(model_config, feature_configs, features, labels, is_training)
    own = self._take_config('multi_tower')
    if len(own.din_towers) or len(own.bst_towers):
      raise AssertionError('sequence towers: use model_class MultiTowerDIN')

  def build_predict_graph(self):
    towers = list(self._model_config.towers)
    inputs = [self._group(t.input)[0] for t in towers]  # (all input-layer calls first, in tower order: :30-35)
    outs = []
    for t, x in zip(towers, inputs):
      x = dnn.batch_norm(x, t.input + '_fea_bn', self._is_training)
      outs.append(self._dnn(x, t.dnn, t.input + '_dnn'))
    top = self._dnn(torch.cat(outs, dim=1), self._model_config.final_dnn, 'final_dnn')
    return self._emit(dnn.dense(top, self._num_class, 'output', head=True))
