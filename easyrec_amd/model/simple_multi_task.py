"""SimpleMultiTask (reference easy_rec/python/model/simple_multi_task.py:14-54): the shared `all` group feeds one tower
DNN per task, each followed by dense(num_class) named `dnn_output_<i>`; losses / metrics per tower come from
MultiTaskModel (model/multi_task_model.py)."""
from easyrec_amd.layers import dnn
from easyrec_amd.model.multi_task_model import MultiTaskModel
from easyrec_amd.protos.simple_multi_task_pb2 import SimpleMultiTask as SimpleMultiTaskConfig


class SimpleMultiTask(MultiTaskModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(SimpleMultiTask, self).__init__(model_config, feature_configs, features, labels, is_training)
    assert self._model_config.WhichOneof('model') == 'simple_multi_task', \
        'invalid model config: %s' % self._model_config.WhichOneof('model')
    self._model_config = self._model_config.simple_multi_task
    assert isinstance(self._model_config, SimpleMultiTaskConfig)
    assert not self.has_backbone, 'SimpleMultiTask over a backbone: see layers/backbone.py'
    self._init_towers(self._model_config.task_towers)

  def build_predict_graph(self):
    self._features, _ = self._input_layer(self._feature_dict, 'all')
    tower_outputs = {}
    for i, task_tower_cfg in enumerate(self._model_config.task_towers):
      tower_name = task_tower_cfg.tower_name
      task_dnn = dnn.DNN(task_tower_cfg.dnn, self._l2_reg, name=tower_name, is_training=self._is_training)
      task_fea = task_dnn(self._features)
      tower_outputs[tower_name] = dnn.dense(task_fea, task_tower_cfg.num_class, 'dnn_output_%d' % i,
                                            l2_reg=self._l2_reg)
    self._add_to_prediction_dict(tower_outputs)
    return self._prediction_dict
