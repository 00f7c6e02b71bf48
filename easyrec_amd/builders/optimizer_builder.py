"""Optimizer + learning-rate schedule from `train_config.optimizer_config`.

Mirror of reference easy_rec/python/builders/optimizer_builder.py:28-211 and
core/learning_schedules.py:25-73.  The reference builds TF optimizer objects; here `build` returns a
small host-side state machine that produces, per step, the fp32 scalars of `er_opt_hyper`
(include/easyrec_hip.h) exactly the way TF computes them on its side:
  tf.train.AdamOptimizer: beta1_power/beta2_power are fp32 variables starting at beta and multiplied
  by beta after every apply; lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power); epsilon 1e-8.
  tf.train.exponential_decay: lr * rate^(step/decay_steps) (floor when staircase), evaluated with the
  pre-increment global step.
"""
import math

import numpy as np

from easyrec_amd import kernels

F32 = np.float32


def exponential_decay_with_burnin(global_step, learning_rate_base, decay_steps, decay_factor,
                                  burnin_learning_rate=0.0, burnin_steps=0, min_learning_rate=0.0,
                                  staircase=True):
  """reference core/learning_schedules.py:25-73 (fp32 arithmetic as in the TF graph)."""
  if burnin_learning_rate == 0:
    burnin_rate = F32(learning_rate_base)
  else:
    slope = (learning_rate_base - burnin_learning_rate) / burnin_steps
    burnin_rate = F32(slope) * F32(global_step) + F32(burnin_learning_rate)
  p = F32(global_step - burnin_steps) / F32(decay_steps)
  if staircase:
    p = np.floor(p)
  post = F32(learning_rate_base) * np.power(F32(decay_factor), F32(p), dtype=F32)
  lr = burnin_rate if global_step < burnin_steps else post
  return F32(max(F32(lr), F32(min_learning_rate)))


def _make_schedule(lr_config):
  kind = lr_config.WhichOneof('learning_rate')
  if kind == 'constant_learning_rate':
    c = lr_config.constant_learning_rate
    return lambda step: F32(c.learning_rate)
  if kind == 'exponential_decay_learning_rate':
    c = lr_config.exponential_decay_learning_rate
    return lambda step: exponential_decay_with_burnin(
        step, c.initial_learning_rate, c.decay_steps, c.decay_factor,
        burnin_learning_rate=c.burnin_learning_rate, burnin_steps=c.burnin_steps,
        min_learning_rate=c.min_learning_rate, staircase=c.staircase)
  if kind == 'manual_step_learning_rate':
    c = lr_config.manual_step_learning_rate
    if not c.schedule:
      raise ValueError('Empty learning rate schedule.')
    bounds = [x.step for x in c.schedule]
    rates = [c.initial_learning_rate] + [x.learning_rate for x in c.schedule]

    def manual(step):
      idx = sum(1 for b in bounds if step >= b)
      if c.warmup and idx == 0 and bounds:
        slope = (rates[1] - rates[0]) / float(bounds[0])
        return F32(rates[0] + slope * step)
      return F32(rates[idx])

    return manual
  if kind == 'cosine_decay_learning_rate':
    c = lr_config.cosine_decay_learning_rate

    def cosine(step):
      # fp32 throughout, in the graph's order (core/learning_schedules.py:113-127: python floats are constants of the
      # tensor's type): near the end of the decay 1 + cos(x) keeps only a few bits, and they should be the same bits
      x = F32(math.pi) * (F32(step) - F32(c.warmup_steps) - F32(c.hold_base_rate_steps)) / \
          F32(float(c.total_steps - c.warmup_steps - c.hold_base_rate_steps))
      lr = F32(0.5 * c.learning_rate_base) * (F32(1.0) + np.cos(x, dtype=np.float32))
      if c.hold_base_rate_steps > 0 and step <= c.warmup_steps + c.hold_base_rate_steps:
        lr = F32(c.learning_rate_base)
      if c.warmup_steps > 0 and step < c.warmup_steps:
        slope = (c.learning_rate_base - c.warmup_learning_rate) / c.warmup_steps
        lr = F32(slope) * F32(step) + F32(c.warmup_learning_rate)
      return F32(0.0 if step > c.total_steps else lr)

    return cosine
  if kind == 'poly_decay_learning_rate':
    c = lr_config.poly_decay_learning_rate

    def poly(step):
      s = min(step, c.total_steps)
      return F32((c.learning_rate_base - c.end_learning_rate) * (1 - s / float(c.total_steps))**c.power +
                 c.end_learning_rate)

    return poly
  raise ValueError('Learning_rate %s not supported.' % kind)


class OptimizerState(object):
  """Host-side twin of a TF optimizer's non-slot state; fills er_opt_hyper rows."""

  def __init__(self, kind, schedule, beta1=0.9, beta2=0.999, epsilon=1e-8, name=''):
    self.kind = kind
    self.schedule = schedule
    self.name = name
    self.beta1, self.beta2, self.epsilon = F32(beta1), F32(beta2), F32(epsilon)
    self.beta1_power, self.beta2_power = F32(beta1), F32(beta2)

  def hyper_row(self, global_step, grad_scale=1.0):
    """fp32 scalars for the update that uses the pre-increment `global_step`."""
    row = np.zeros(kernels.HYPER_FLOATS, dtype=np.float32)
    lr = F32(self.schedule(global_step))
    row[kernels.HYPER_LR] = lr
    if self.kind in (kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM):
      one = F32(1.0)
      lr_t = lr * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
      row[kernels.HYPER_LR_T] = F32(lr_t)
      row[kernels.HYPER_BETA1] = self.beta1
      row[kernels.HYPER_BETA2] = self.beta2
      row[kernels.HYPER_OMB1] = one - self.beta1
      row[kernels.HYPER_OMB2] = one - self.beta2
      row[kernels.HYPER_EPS] = self.epsilon
    row[kernels.HYPER_GSCALE] = F32(grad_scale)
    return row

  def reset_to_step(self, step):
    """State after `step` applies, e.g. when a checkpoint is restored: TF keeps beta1_power / beta2_power as fp32
    variables multiplied once per apply, so they are re-derived by the same sequence of fp32 multiplications."""
    if self.kind in (kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM):
      self.beta1_power, self.beta2_power = F32(self.beta1), F32(self.beta2)
      for _ in range(int(step)):
        self.finish_step()

  def finish_step(self):
    """AdamOptimizer._finish: beta powers advance after the apply."""
    if self.kind in (kernels.OPT_ADAM, kernels.OPT_LAZY_ADAM):
      self.beta1_power = F32(self.beta1_power * self.beta1)
      self.beta2_power = F32(self.beta2_power * self.beta2)


def build(optimizer_config):
  """Returns an `OptimizerState` (reference optimizer_builder.build :28-144)."""
  opt_type = optimizer_config.WhichOneof('optimizer')
  if opt_type == 'adam_optimizer':
    c = optimizer_config.adam_optimizer
    return OptimizerState(kernels.OPT_ADAM, _make_schedule(c.learning_rate), c.beta1, c.beta2, name=opt_type)
  if opt_type == 'lazy_adam_optimizer':
    c = optimizer_config.lazy_adam_optimizer
    return OptimizerState(kernels.OPT_LAZY_ADAM, _make_schedule(c.learning_rate), c.beta1, c.beta2,
                          name=opt_type)
  if opt_type == 'adagrad_optimizer':
    c = optimizer_config.adagrad_optimizer
    st = OptimizerState(kernels.OPT_ADAGRAD, _make_schedule(c.learning_rate), name=opt_type)
    st.initial_accumulator_value = c.initial_accumulator_value
    return st
  if opt_type == 'momentum_optimizer' and optimizer_config.momentum_optimizer.momentum_optimizer_value == 0:
    c = optimizer_config.momentum_optimizer
    return OptimizerState(kernels.OPT_SGD, _make_schedule(c.learning_rate), name=opt_type)
  raise ValueError('Optimizer %s not supported on the MI355X path (supported: adam_optimizer, '
                   'lazy_adam_optimizer, adagrad_optimizer, momentum_optimizer with momentum 0).' % opt_type)
