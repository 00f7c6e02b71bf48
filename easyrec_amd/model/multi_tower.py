"""MultiTower (reference easy_rec/python/model/multi_tower.py:15-62): per feature group a BatchNorm + DNN tower,
final_dnn over the concatenated tower outputs, dense(num_class) WITHOUT kernel regulariser (:58)."""
import torch

from easyrec_amd.layers import dnn
from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos.multi_tower_pb2 import MultiTower as MultiTowerConfig


class MultiTower(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(MultiTower, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert self._model_config.WhichOneof('model') == 'multi_tower', \
        'invalid model config: %s' % self._model_config.WhichOneof('model')
    self._model_config = self._model_config.multi_tower
    assert isinstance(self._model_config, MultiTowerConfig)
    assert len(self._model_config.din_towers) == 0 and len(self._model_config.bst_towers) == 0, \
        'sequence towers: use model_class MultiTowerDIN'
    self._tower_num = len(self._model_config.towers)

  def build_predict_graph(self):
    tower_features = [self._input_layer(self._feature_dict, tower.input)[0] for tower in self._model_config.towers]
    tower_fea_arr = []
    for tower, tower_fea in zip(self._model_config.towers, tower_features):
      tower_name = tower.input
      tower_fea = dnn.batch_norm(tower_fea, '%s_fea_bn' % tower_name, self._is_training)
      tower_dnn_layer = dnn.DNN(tower.dnn, self._l2_reg, '%s_dnn' % tower_name, self._is_training)
      tower_fea_arr.append(tower_dnn_layer(tower_fea))
    all_fea = torch.cat(tower_fea_arr, dim=1)
    final_dnn_layer = dnn.DNN(self._model_config.final_dnn, self._l2_reg, 'final_dnn', self._is_training)
    all_fea = final_dnn_layer(all_fea)
    output = dnn.dense(all_fea, self._num_class, 'output')
    self._add_to_prediction_dict(output)
    return self._prediction_dict
