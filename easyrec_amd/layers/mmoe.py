"""MMOE layer (reference easy_rec/python/layers/mmoe.py:13-83).

E expert DNNs over the shared input; per task a softmax gate `dense_E(x)` mixes them.  The reference
stacks the experts ([B,E,H] copy) and runs Softmax/Mul/Sum per task; here ONE launch (`er_mmoe_mix`)
soft-maxes all T gates and writes all T mixtures.
"""
import logging

import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn


class MMOE(object):

  def __init__(self, expert_dnn_config, l2_reg, num_task, num_expert=None, name='mmoe', is_training=False):
    if isinstance(expert_dnn_config, list):
      self._expert_dnn_configs = expert_dnn_config
      self._num_expert = len(expert_dnn_config)
    else:
      assert num_expert is not None and num_expert > 0, \
          'param `num_expert` must be large than zero, when expert_dnn_config is not a list'
      self._expert_dnn_configs = [expert_dnn_config] * num_expert
      self._num_expert = num_expert
    logging.info('num_expert: {0}'.format(self._num_expert))
    self._num_task = num_task
    self._l2_reg = l2_reg
    self._name = name
    self._is_training = is_training

  @property
  def num_expert(self):
    return self._num_expert

  def __call__(self, deep_fea):
    expert_fea_list = []
    for expert_id in range(self._num_expert):
      expert_dnn = dnn.DNN(self._expert_dnn_configs[expert_id], self._l2_reg,
                           name='%s/expert_%d' % (self._name, expert_id), is_training=self._is_training)
      expert_fea_list.append(expert_dnn(deep_fea))
    experts = torch.stack(expert_fea_list, dim=0)  # [E, B, H]
    gate_logits = torch.stack([
        dnn.dense(deep_fea, self._num_expert, '%s/gate_%d/dnn' % (self._name, task_id), l2_reg=self._l2_reg)
        for task_id in range(self._num_task)
    ], dim=0)  # [T, B, E]
    mixed = kernels.MMoEMixFn.apply(experts, gate_logits)  # [T, B, H]
    return [mixed[t] for t in range(self._num_task)]
