"""MMoE (reference easy_rec/python/model/mmoe.py:14-70): shared `all` group -> MMOE layer -> per-task
tower DNN -> dense(num_class) named `dnn_output_<i>`."""
from easyrec_amd.layers import mmoe
from easyrec_amd.model.multi_task_model import MultiTaskModel


class MMoE(MultiTaskModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(MMoE, self).__init__(model_config, feature_configs, features, labels, is_training)
    own = self._take_config('mmoe')
    if self.has_backbone:
      raise AssertionError('MMoE over a backbone: see layers/backbone.py')
    self._init_towers(own.task_towers)

  def build_predict_graph(self):
    own = self._model_config
    self._features = self._group('all')[0]
    # The reference builds the layer WITHOUT is_training (model/mmoe.py:37-47; layers/mmoe.py:14-20 defaults it to
    # False): the experts' BatchNorm normalises with the moving statistics - zeros / ones unless a checkpoint says
    # otherwise, never updated - and their dropout is off, in training as in evaluation.  Kept, so that losses,
    # gradients and checkpoints match the reference's (pinned by tests/test_reference_layers.py's model assemblies).
    if own.HasField('expert_dnn'):
      layer = mmoe.MMOE(own.expert_dnn, l2_reg=self._l2_reg, num_task=self._task_num, num_expert=own.num_expert)
    else:  # the older config form: a list of named experts
      layer = mmoe.MMOE([e.dnn for e in own.experts], l2_reg=self._l2_reg, num_task=self._task_num)
    return self._tower_heads(layer(self._features))
