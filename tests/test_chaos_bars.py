"""CPU: what justifies the loss bars of the GPU trajectory tests (tests/_bars.py).

The oracle (oracle/model_oracle.py OracleTrainer: the CPU restatement of the reference's training step) is run twice from
the same state over the same batches; the twin's gradients are perturbed, before the optimizer sees them, by one fp32 ulp of
each tensor's largest element with random signs (`grad_noise`, rows of an embedding table that no id touched stay exactly
zero): a model of ANOTHER fp32 summation order - which is all that separates the HIP kernels from the oracle.  The largest
relative loss difference per step over a grid of model seeds x noise seeds is the chaos envelope D[step] of the config.

A trajectory bar is JUSTIFIED when it does not exceed max(1e-4, 10 * D[step]): the GPU may deviate from the oracle by no
more than ten times what a one-ulp change of the gradients alone produces (tools/chaos_probe.py and
profiles/r05_s19_din_small_chaos_probe.txt were the notes this test replaces).  The test also shows the amplification
itself: a perturbation of one ulp (6e-8) is more than a thousand ulp of the loss by the last step."""
import os

import numpy as np
import pytest

import _bars
from easyrec_amd.utils import config_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ULP = 2.0 ** -24


def _envelope(name, B, steps, model_seeds, noise_seeds, gen_seed, criteo=False, kv=False):
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  env = np.zeros(steps)
  for ms in model_seeds:
    cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', name))
    if kv:
      for f in cfg.feature_config.features:
        if f.feature_type == f.SequenceFeature:
          f.ev_params.max_capacity = 2048
    est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=ms).build()
    state0 = est.state_dict()
    gen = (SyntheticCriteo if criteo else SyntheticBatches)(cfg.data_config, est.feature_configs, batch_size=B, seed=gen_seed)
    batches = [gen.next_batch() for _ in range(steps)]
    base = OracleTrainer(cfg, state0, batch_size=B)
    ref = [base.train_step(b) for b in batches]
    for ns in noise_seeds:
      rng = np.random.default_rng(1000 * ms + ns)
      twin = OracleTrainer(cfg, state0, batch_size=B)

      def noise(n, g, rng=rng):
        noisy = g + np.float32(ULP * float(np.abs(g).max())) * rng.choice(np.float32([-1.0, 1.0]), size=g.shape)
        if n.endswith('/embedding_weights'):
          noisy = np.where(np.abs(g).sum(axis=-1, keepdims=True) > 0, noisy, g)
        return noisy

      twin.grad_noise = noise
      for i, b in enumerate(batches):
        out = twin.train_step(b)
        env[i] = max(env[i], max(abs(out[k] - ref[i][k]) / max(1.0, abs(ref[i][k])) for k in ref[i]))
  return env


CASES = {
    'deepfm_criteo_small': dict(name='deepfm_criteo_small.config', B=256, steps=5, model_seeds=(3, 4, 5, 6, 7, 8),
                                noise_seeds=(0, 1, 2), gen_seed=3, criteo=True),
    'din_taobao_small': dict(name='din_taobao_small.config', B=48, steps=3, model_seeds=(4, 5, 6, 7, 8, 9),
                             noise_seeds=(0, 1, 2), gen_seed=12, kv=True),
}


@pytest.mark.parametrize('config', sorted(CASES))
def test_trajectory_bars_are_justified_by_the_measured_one_ulp_envelope(ref_backend, config):
  env = _envelope(**CASES[config])
  bars = _bars.TRAJECTORY[config]
  assert len(bars) == len(env)
  assert env[0] == 0.0, 'the first step runs from identical parameters: nothing to amplify yet'
  assert bars[0] == _bars.FLOOR
  running = np.maximum.accumulate(env)
  for step in range(1, len(env)):
    allowed = max(_bars.FLOOR, _bars.SLACK * running[step])
    assert bars[step] <= allowed, ('bar looser than %g x the one-ulp envelope' % _bars.SLACK, config, step, bars[step],
                                   list(env))
  # the amplification the bars are there for: one ulp (6e-8) on the gradients is >= 1000 ulp on the loss by the last step
  assert running[-1] >= 1e3 * ULP, list(env)
