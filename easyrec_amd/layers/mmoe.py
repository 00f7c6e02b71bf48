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
  """`expert_dnn_config`: one DNN config shared by `num_expert` experts, or a list of configs (one expert each).
  `is_training` defaults to False as in the reference - whose model classes rely on that default (model/mmoe.py)."""

  def __init__(self, expert_dnn_config, l2_reg, num_task, num_expert=None, name='mmoe', is_training=False):
    if isinstance(expert_dnn_config, list):
      self._expert_configs = list(expert_dnn_config)
    else:
      if not num_expert or num_expert <= 0:
        raise AssertionError('param `num_expert` must be large than zero, when expert_dnn_config is not a list')
      self._expert_configs = [expert_dnn_config] * num_expert
    logging.info('num_expert: {0}'.format(len(self._expert_configs)))
    self._num_task, self._l2_reg, self._name, self._is_training = num_task, l2_reg, name, is_training

  num_expert = property(lambda self: len(self._expert_configs))

  def __call__(self, deep_fea):
    scope, n = self._name, len(self._expert_configs)
    stacks = [dnn.DNN(cfg, self._l2_reg, name='%s/expert_%d' % (scope, e), is_training=self._is_training)
              for e, cfg in enumerate(self._expert_configs)]
    # experts layer by layer (one grouped launch per depth), the gates' projections in the first depth's launch
    outs, gate_logits = dnn.run_parallel(
        stacks, [deep_fea] * n,
        extra_dense=[(deep_fea, n, '%s/gate_%d/dnn' % (scope, t), self._l2_reg) for t in range(self._num_task)])
    if torch.is_grad_enabled() and hasattr(kernels.hip(), 'copy_multi'):
      return list(kernels.MMoEMixManyFn.apply(n, self._num_task, *outs, *gate_logits))  # T mixtures [B, H]
    experts = torch.stack(outs, dim=0)  # [E, B, H]
    gates = torch.stack(gate_logits, dim=0)  # [T, B, E] logits
    mixed = kernels.MMoEMixFn.apply(experts, gates)  # [T, B, H]
    return [mixed[t] for t in range(self._num_task)]
