#!/usr/bin/env python
"""Which optimizer, with which arguments and which learning rate at which step, the REFERENCE'S OWN
builders/optimizer_builder.py (:28-211) and core/learning_schedules.py build from optimizer configs - run in the build
container where /root/reference exists.  tf.train.*Optimizer / AdamOptimizerS constructors are RECORDED; the schedules
run on a numpy stand-in for the ops they call (cos, where, one_hot, reduce_max ...; tf.train.exponential_decay and
polynomial_decay by their documented formulas) with the global step set by this script.  The fixture
(tests/golden/optimizer_vectors.json) holds per case the config text, the recorded class + arguments and the learning
rate at a list of steps; tests/test_optimizer_pins.py builds the same from easyrec_amd/builders/optimizer_builder.py.

usage: python tests/golden/make_optimizer_vectors.py [/root/reference]
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

STEPS = [0, 1, 2, 5, 9, 10, 11, 49, 50, 51, 99, 100, 101, 499, 500, 999, 1000, 1001, 2500, 10000, 25000]

CASES = {
    'adam_constant': "adam_optimizer { learning_rate { constant_learning_rate { learning_rate: 0.002 } } beta1: 0.8 beta2: 0.99 }",
    'adam_defaults_exp': "adam_optimizer { learning_rate { exponential_decay_learning_rate { initial_learning_rate: 0.001 "
                         "decay_steps: 1000 decay_factor: 0.5 min_learning_rate: 0.00001 } } }",
    'lazy_adam_burnin': "lazy_adam_optimizer { learning_rate { exponential_decay_learning_rate { initial_learning_rate: 0.01 "
                        "decay_steps: 500 decay_factor: 0.7 burnin_learning_rate: 0.001 burnin_steps: 10 staircase: false } } }",
    'adagrad_manual': "adagrad_optimizer { learning_rate { manual_step_learning_rate { initial_learning_rate: 0.1 "
                      "schedule { step: 10 learning_rate: 0.05 } schedule { step: 100 learning_rate: 0.01 } "
                      "schedule { step: 1000 learning_rate: 0.001 } } } initial_accumulator_value: 0.3 }",
    'adagrad_manual_warmup': "adagrad_optimizer { learning_rate { manual_step_learning_rate { initial_learning_rate: 0.01 "
                             "schedule { step: 50 learning_rate: 0.1 } schedule { step: 500 learning_rate: 0.02 } warmup: true } } }",
    'sgd_cosine': "momentum_optimizer { learning_rate { cosine_decay_learning_rate { learning_rate_base: 0.05 total_steps: 1000 "
                  "warmup_learning_rate: 0.005 warmup_steps: 50 hold_base_rate_steps: 50 } } momentum_optimizer_value: 0.0 }",
    'adam_cosine_plain': "adam_optimizer { learning_rate { cosine_decay_learning_rate { learning_rate_base: 0.01 total_steps: 2500 warmup_learning_rate: 0.0 warmup_steps: 0 } } }",
    'adam_poly': "adam_optimizer { learning_rate { poly_decay_learning_rate { learning_rate_base: 0.01 total_steps: 1000 "
                 "end_learning_rate: 0.0001 power: 2.0 } } }",
}


def main():
  from google.protobuf import text_format

  from easyrec_amd import protos
  STEP = [0]
  recorded = []
  tf = types.ModuleType('tensorflow')
  tf.__version__ = '1.15.0'
  tf.float32, tf.int32, tf.int64 = np.float32, np.int32, np.int64  # (the schedules compute in float32, as the graph does)
  f32 = np.float32
  tf.constant = lambda v, dtype=None, name=None: np.asarray(v, dtype=dtype or np.float32)
  tf.cast = lambda x, dtype: np.asarray(x).astype(dtype)
  tf.cos = lambda x: np.cos(np.asarray(x, dtype=np.float32))
  tf.where = lambda c, a, b, name=None: np.where(c, np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32))
  tf.greater_equal = lambda a, b: np.asarray(a) >= np.asarray(b)
  tf.less = lambda a, b: np.asarray(a) < np.asarray(b)
  tf.maximum = lambda a, b, name=None: np.maximum(np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32))
  tf.reduce_max = lambda x: np.max(x)
  tf.reduce_sum = lambda x, name=None: np.sum(np.asarray(x, dtype=np.float32), dtype=np.float32)
  tf.one_hot = lambda i, depth: np.eye(depth, dtype=np.float32)[int(i)]

  def exponential_decay(lr, step, decay_steps, decay_rate, staircase=False, name=None):
    p = f32(step) / f32(decay_steps)
    if staircase:
      p = np.floor(p)
    return f32(lr) * np.power(f32(decay_rate), p, dtype=np.float32)

  def polynomial_decay(lr, step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False, name=None):
    s = f32(min(step, decay_steps))
    return (f32(lr) - f32(end_learning_rate)) * np.power(f32(1) - s / f32(decay_steps), f32(power), dtype=np.float32) + \
        f32(end_learning_rate)

  def optimizer(cls):
    def make(*args, **kw):
      names = {'AdagradOptimizer': ['learning_rate'], 'AdamOptimizer': ['learning_rate'], 'MomentumOptimizer': ['learning_rate'],
               'AdamOptimizerS': ['learning_rate']}[cls]
      kw.update(dict(zip(names, args)))
      kw.pop('learning_rate')
      recorded.append((cls, {k: float(v) for k, v in kw.items()}))
      return cls
    return make

  tf.train = types.SimpleNamespace(exponential_decay=exponential_decay, polynomial_decay=polynomial_decay,
                                   get_or_create_global_step=lambda: np.int64(STEP[0]),
                                   AdamOptimizer=optimizer('AdamOptimizer'), AdagradOptimizer=optimizer('AdagradOptimizer'),
                                   MomentumOptimizer=optimizer('MomentumOptimizer'))
  tf.compat = types.SimpleNamespace(v1=tf)
  sys.modules['tensorflow'] = tf
  for name in ('easy_rec', 'easy_rec.python', 'easy_rec.python.compat', 'easy_rec.python.compat.weight_decay_optimizers',
               'easy_rec.python.compat.adam_s', 'easy_rec.python.core'):
    sys.modules[name] = types.ModuleType(name)
  sys.modules['easy_rec.python.compat'].weight_decay_optimizers = sys.modules['easy_rec.python.compat.weight_decay_optimizers']
  sys.modules['easy_rec.python.compat.adam_s'].AdamOptimizerS = optimizer('AdamOptimizerS')

  def load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sys.modules[name] = m
    return m

  sched = load('easy_rec/python/core/learning_schedules.py', 'easy_rec.python.core.learning_schedules')
  sys.modules['easy_rec.python.core'].learning_schedules = sched
  ob = load('easy_rec/python/builders/optimizer_builder.py', 'ref_optimizer_builder')
  cases = []
  for tag, text in CASES.items():
    cfg = protos.optimizer_pb2.Optimizer()
    text_format.Merge(text, cfg)
    lrs = []
    for s in STEPS:
      STEP[0] = s
      del recorded[:]
      _, summary = ob.build(cfg)
      lrs.append(float(np.asarray(summary[0], dtype=np.float32)))
    cls, kwargs = recorded[0]
    cases.append({'tag': tag, 'config': text, 'class': cls, 'kwargs': kwargs, 'steps': STEPS, 'learning_rate': lrs})
  path = os.path.join(HERE, 'optimizer_vectors.json')
  with open(path, 'w') as f:
    json.dump({'generator': 'tests/golden/make_optimizer_vectors.py', 'cases': cases}, f, indent=1, sort_keys=True)
  print('wrote %s: %d cases' % (path, len(cases)))
  for c in cases:
    print(' ', c['tag'], c['class'], c['kwargs'], ['%.6g' % v for v in c['learning_rate'][:4]], '...')


if __name__ == '__main__':
  main()
