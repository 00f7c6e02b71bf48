"""MultiTaskModel (reference easy_rec/python/model/multi_task_model.py:19-300).

Per task tower: predictions with suffix `_<tower_name>` (:124-141), loss = task weight x loss of the
tower's label (:200-240), summed into total_loss by the estimator.  Fixed loss-weight strategy only.
"""
import logging
from collections import OrderedDict

from easyrec_amd.model.rank_model import RankModel
from easyrec_amd.protos import tower_pb2


class MultiTaskModel(RankModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(MultiTaskModel, self).__init__(model_config, feature_configs, features, labels, is_training)
    self._task_towers = []
    self._task_num = None
    self._label_name_dict = {}

  def _init_towers(self, task_tower_configs):
    self._task_towers = task_tower_configs
    self._task_num = len(task_tower_configs)
    for i, task_tower_config in enumerate(task_tower_configs):
      assert isinstance(task_tower_config, (tower_pb2.TaskTower, tower_pb2.BayesTaskTower)), \
          'task_tower_config must be a instance of tower_pb2.TaskTower or tower_pb2.BayesTaskTower'
      tower_name = task_tower_config.tower_name
      if self._labels is not None:
        if task_tower_config.HasField('label_name'):
          label_name = task_tower_config.label_name
        else:
          label_name = list(self._labels.keys())[i]
          logging.info('Task Tower [%s] use label [%s]' % (tower_name, label_name))
        assert label_name in self._labels, 'label [%s] must exists in labels' % label_name
        self._label_name_dict[tower_name] = label_name

  def _tower_heads(self, inputs_per_task):
    """Shared tail of the multi-task models here (MMoE, SimpleMultiTask): task t's input goes through the tower's
    DNN when it has one, then the `dnn_output_<t>` projection to num_class (reference model/mmoe.py:56-68,
    model/simple_multi_task.py:38-52); fills the prediction dict."""
    from easyrec_amd.layers import dnn
    heads = {}
    for t, tower in enumerate(self._task_towers):
      h = inputs_per_task[t]
      if tower.HasField('dnn'):
        h = dnn.DNN(tower.dnn, self._l2_reg, name=tower.tower_name, is_training=self._is_training)(h)
      heads[tower.tower_name] = dnn.dense(h, tower.num_class, 'dnn_output_%d' % t, l2_reg=self._l2_reg)
    self._add_to_prediction_dict(heads)
    return self._prediction_dict

  def _add_to_prediction_dict(self, output):
    for task_tower_cfg in self._task_towers:
      tower_name = task_tower_cfg.tower_name
      assert len(task_tower_cfg.losses) == 0, 'per-tower `losses` lists are outside the hot-path scope'
      self._prediction_dict.update(
          self._output_to_prediction_impl(output[tower_name], loss_type=task_tower_cfg.loss_type,
                                          num_class=task_tower_cfg.num_class, suffix='_%s' % tower_name))

  def build_loss_weight(self):
    loss_weights = OrderedDict()
    for task_tower_cfg in self._task_towers:
      loss_weights[task_tower_cfg.tower_name] = [task_tower_cfg.weight]
    strategy = self._base_model_config.loss_weight_strategy
    assert strategy == self._base_model_config.Fixed, 'only the Fixed loss weight strategy is supported'
    return loss_weights

  def build_loss_graph(self):
    task_loss_weights = self.build_loss_weight()
    for task_tower_cfg in self._task_towers:
      tower_name = task_tower_cfg.tower_name
      loss_weight = 1.0
      if task_tower_cfg.use_sample_weight:
        loss_weight = self._sample_weight
      assert not task_tower_cfg.HasField('task_space_indicator_label') and \
          not task_tower_cfg.HasField('task_space_indicator_name'), 'task-space weighting: outside the hot-path scope'
      # the task weight multiplies the loss AND its gradient inside the fused loss kernel
      loss_dict = self._build_loss_impl(task_tower_cfg.loss_type, label_name=self._label_name_dict[tower_name],
                                        loss_weight=loss_weight, num_class=task_tower_cfg.num_class,
                                        suffix='_%s' % tower_name, loss_scale=task_loss_weights[tower_name][0])
      self._loss_dict.update(loss_dict)
    return self._loss_dict

  def get_outputs(self):
    outputs = []
    for task_tower_cfg in self._task_towers:
      outputs.extend(self._get_outputs_impl(task_tower_cfg.loss_type, task_tower_cfg.num_class,
                                            suffix='_%s' % task_tower_cfg.tower_name))
    return list(set(outputs))
