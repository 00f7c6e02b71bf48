"""Keras-style MMoE layer of the backbone API (reference easy_rec/python/layers/keras/multi_task.py:14-68):
`num_expert` expert MLPs over the input, per task a softmax gate `Dense(num_expert)`; returns the list of
task inputs.  The gate softmax + mixture of all tasks is ONE launch (er_mmoe_mix)."""
import torch

from easyrec_amd import kernels
from easyrec_amd.layers import dnn
from easyrec_amd.layers.keras.blocks import MLP


class MMoE(object):

  def __init__(self, params, name='MMoE', reuse=None, **kwargs):
    self.name = name
    params.check_required(['num_expert', 'num_task'])
    self._num_expert = int(params.num_expert)
    self._num_task = int(params.num_task)
    self._l2_reg = params.l2_regularizer
    self._experts = []
    if params.has_field('expert_mlp'):
      expert_params = params.expert_mlp
      expert_params.l2_regularizer = self._l2_reg
      # (variables of the nested layers carry this layer's name scope too: `<block>/expert_i/...`, `<block>/gate_t/...`)
      self._experts = [MLP(expert_params, '%s/expert_%d' % (name, i), reuse) for i in range(self._num_expert)]

  def __call__(self, inputs, training=None, **kwargs):
    if self._num_expert == 0:
      return inputs
    if self._experts:
      expert_fea_list = [e(inputs, training=training) for e in self._experts]
    else:
      expert_fea_list = list(inputs)[:self._num_expert]
    experts = torch.stack(expert_fea_list, dim=0)
    # without built-in expert MLPs the gate reads the extra last input (multi_task.py:57-58)
    gate_input = inputs if self._experts else inputs[self._num_expert]
    gate_logits = torch.stack([
        dnn.dense(gate_input, self._num_expert, '%s/gate_%d' % (self.name, t), l2_reg=self._l2_reg)
        for t in range(self._num_task)
    ], dim=0)
    mixed = kernels.MMoEMixFn.apply(experts, gate_logits)
    return [mixed[t] for t in range(self._num_task)]
