"""DBMTL - deep Bayesian multi-task learning (reference easy_rec/python/model/dbmtl.py:17-116).

bottom: the `all` group (optionally through `bottom_dnn`); per task a tower DNN (`<tower>/dnn`), then the Bayesian
chain: `<tower>/relation_dnn` over [the tower's features, the relation features of the towers named in
relation_tower_names (which must come earlier)], and the `<tower>/output` projection.  With `expert_dnn` an MMoE block
sits between the bottom and the towers (dbmtl.py:64-72; built, as in the reference, without is_training: its experts'
BatchNorm runs on the moving statistics).  The CMBF / Uniter bottoms are outside the hot-path scope."""
import torch

from easyrec_amd.layers import dnn
from easyrec_amd.layers import mmoe
from easyrec_amd.model.multi_task_model import MultiTaskModel
from easyrec_amd.protos.dbmtl_pb2 import DBMTL as DBMTLConfig


class DBMTL(MultiTaskModel):

  def __init__(self, model_config, feature_configs, features, labels=None, is_training=False):
    super(DBMTL, self).__init__(model_config, feature_configs, features, labels, is_training)
    kind = self._model_config.WhichOneof('model')
    assert kind == 'dbmtl', 'invalid model config: %s' % kind
    self._model_config = self._model_config.dbmtl
    assert isinstance(self._model_config, DBMTLConfig)
    for field in ('bottom_cmbf', 'bottom_uniter'):
      if self._model_config.HasField(field):
        raise NotImplementedError('DBMTL.%s is outside the hot-path scope' % field)
    assert not self.has_backbone, 'DBMTL over a backbone: see layers/backbone.py'
    self._init_towers(self._model_config.task_towers)

  def build_predict_graph(self):
    c = self._model_config
    bottom, _ = self._input_layer(self._feature_dict, 'all')
    self._features = bottom
    if c.HasField('bottom_dnn'):
      bottom = dnn.DNN(c.bottom_dnn, self._l2_reg, name='bottom_dnn', is_training=self._is_training)(bottom)
    if c.HasField('expert_dnn'):
      task_inputs = mmoe.MMOE(c.expert_dnn, l2_reg=self._l2_reg, num_task=self._task_num, num_expert=c.num_expert)(bottom)
    else:
      task_inputs = [bottom] * self._task_num
    relation, logits = {}, {}
    for tower, own in zip(c.task_towers, task_inputs):
      name = tower.tower_name
      if tower.HasField('dnn'):
        own = dnn.DNN(tower.dnn, self._l2_reg, name=name + '/dnn', is_training=self._is_training)(own)
      parts = [own] + [relation[r] for r in tower.relation_tower_names]
      joined = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
      relation[name] = dnn.DNN(tower.relation_dnn, self._l2_reg, name=name + '/relation_dnn',
                               is_training=self._is_training)(joined)
      logits[name] = dnn.dense(relation[name], tower.num_class, name + '/output', l2_reg=self._l2_reg)
    self._add_to_prediction_dict(logits)
    return self._prediction_dict
