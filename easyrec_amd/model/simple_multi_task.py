"""SimpleMultiTask (reference easy_rec/python/model/simple_multi_task.py:14-54): every task's tower reads the same
`all` group; tower DNN + `dnn_output_<i>` projection, losses and metrics per tower are MultiTaskModel's."""
from easyrec_amd.model.multi_task_model import MultiTaskModel
from easyrec_amd.protos.simple_multi_task_pb2 import SimpleMultiTask as SimpleMultiTaskConfig


class SimpleMultiTask(MultiTaskModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(SimpleMultiTask, self).__init__(model_config, feature_configs, features, labels, is_training)
    kind = self._model_config.WhichOneof('model')
    assert kind == 'simple_multi_task', 'invalid model config: %s' % kind
    self._model_config = self._model_config.simple_multi_task
    assert isinstance(self._model_config, SimpleMultiTaskConfig)
    assert not self.has_backbone, 'SimpleMultiTask over a backbone: see layers/backbone.py'
    self._init_towers(self._model_config.task_towers)

  def build_predict_graph(self):
    shared, _ = self._input_layer(self._feature_dict, 'all')
    self._features = shared
    return self._tower_heads([shared] * self._task_num)
