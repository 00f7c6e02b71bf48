"""SimpleMultiTask (reference easy_rec/python/model/simple_multi_task.py:14-54): every task's tower reads the same
`all` group; tower DNN + `dnn_output_<i>` projection, losses and metrics per tower are MultiTaskModel's."""
from easyrec_amd.model.multi_task_model import MultiTaskModel


class SimpleMultiTask(MultiTaskModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(SimpleMultiTask, self).__init__(model_config, feature_configs, features, labels, is_training)
    own = self._take_config('simple_multi_task')
    if self.has_backbone:
      raise AssertionError('SimpleMultiTask over a backbone: see layers/backbone.py')
    self._init_towers(own.task_towers)

  def build_predict_graph(self):
    self._features = self._group('all')[0]
    return self._tower_heads([self._features] * self._task_num)
