"""easyrec_amd/builders/optimizer_builder.py against the REFERENCE'S OWN builders/optimizer_builder.py +
core/learning_schedules.py (tests/golden/make_optimizer_vectors.py, run where /root/reference exists): the optimizer a
config selects, its hyper-parameters, and the learning rate at 21 steps of every schedule kind (constant, exponential
decay with staircase / burn-in / floor, manual steps with and without warm-up, cosine with warm-up and hold,
polynomial)."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'optimizer_vectors.json')) as f:
  CASES = {c['tag']: c for c in json.load(f)['cases']}


@pytest.mark.parametrize('tag', sorted(CASES))
def test_optimizer_and_schedule(tag):
  from google.protobuf import text_format

  from easyrec_amd import kernels
  from easyrec_amd.builders import optimizer_builder
  from easyrec_amd.protos import optimizer_pb2
  case = CASES[tag]
  cfg = optimizer_pb2.Optimizer()
  text_format.Merge(case['config'], cfg)
  opt = optimizer_builder.build(cfg)
  want_kind = {'AdamOptimizer': kernels.OPT_ADAM, 'AdamOptimizerS': kernels.OPT_LAZY_ADAM,
               'AdagradOptimizer': kernels.OPT_ADAGRAD, 'MomentumOptimizer': kernels.OPT_SGD}[case['class']]
  assert opt.kind == want_kind
  kw = case['kwargs']
  if 'beta1' in kw:
    assert abs(opt.beta1 - kw['beta1']) < 1e-7 and abs(opt.beta2 - kw['beta2']) < 1e-7
    assert opt.epsilon == 1e-8  # (the builder passes none: TensorFlow's default)
  if 'initial_accumulator_value' in kw:
    assert abs(opt.initial_accumulator_value - kw['initial_accumulator_value']) < 1e-7
  if 'momentum' in kw:
    assert kw['momentum'] == 0.0  # (plain gradient descent is the only momentum value on this path)
  for step, want in zip(case['steps'], case['learning_rate']):
    got = float(opt.schedule(step))
    assert abs(got - want) <= 2e-6 * max(abs(want), 1e-12) + 1e-12, (tag, step, got, want)
