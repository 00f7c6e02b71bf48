"""Test infrastructure shared by tests/test_multi_rank_oracle.py (CPU stand-in backend) and its -m gpu twin: W
embedding-parallel ranks as W threads (tests/_sim_comm.py), each with its OWN batch, against the oracle's W-worker step
(`OracleTrainer.train_step_world`: per-worker BatchNorm statistics, gradients averaged over the workers, row gradients
summed at the owners and divided by W - reference compat/optimizers.py:285-345)."""
import copy
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_cfg(name, lazy=False, clip=0.0):
  from easyrec_amd.utils import config_util
  cfg = copy.deepcopy(config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', name)))
  if lazy:
    oc = cfg.train_config.optimizer_config[0]
    oc.lazy_adam_optimizer.learning_rate.CopyFrom(oc.adam_optimizer.learning_rate)
  if clip:
    cfg.train_config.gradient_clipping_by_norm = clip
  return cfg


def rank_batches(cfg, feature_configs, B, world, steps, seed=100):
  """batches[step][rank]: every rank draws from its own generator."""
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  gens = [SyntheticCriteo(cfg.data_config, feature_configs, batch_size=B, seed=seed + 17 * r) for r in range(world)]
  return [[g.next_batch() for g in gens] for _ in range(steps)]


def run_world(cfg, device, B, world, batches, padded=True, recv_slack=2.0, seed=4, capture=False):
  """-> (per-rank [state_dict(slots), [loss dict per step], grad norms], the W-worker oracle).  The oracle starts from
  rank 0's gathered initial state (identical on every rank by construction: same seed)."""
  import torch
  from _sim_comm import SimWorld
  from easyrec_amd.model.embedding_parallel import EmbeddingParallelEstimator
  from oracle.model_oracle import OracleTrainer
  os.environ['EASYREC_AMD_PADDED_EXCHANGE'] = '1' if padded else '0'
  sim = SimWorld(world)
  init = {}

  def rank_fn(rank, comm):
    if str(device).startswith('cuda'):
      torch.cuda.set_device(0)
    est = EmbeddingParallelEstimator(cfg, device=device, batch_size=B, seed=seed, rank=rank, world=world, comm=comm,
                                     replicate_bytes=1024, recv_slack=recv_slack).build()
    assert est.engine.padded == padded
    st0 = est.state_dict()  # collective
    if rank == 0:
      init.update(st0)
    losses, norms = [], []
    for step_batches in batches:
      est.train_step(step_batches[rank])
      losses.append(est.loss_values())
      norms.append(float(est.grad_norm.item()))
    return est.state_dict(slots=True), losses, norms

  results = sim.run(rank_fn)
  orc = OracleTrainer(cfg, init, batch_size=B)
  exp_losses = [orc.train_step_world(step_batches) for step_batches in batches[:1]]
  orc_first = {k: v.copy() for k, v in orc.slots.items()}
  orc_norm0 = orc.last_grad_norm
  moving0 = [dict(m) for m in orc.rank_moving]
  for step_batches in batches[1:]:
    exp_losses.append(orc.train_step_world(step_batches))
  return results, orc, exp_losses, orc_first, orc_norm0, moving0


def check_against_oracle(results, orc, exp_losses, orc_first, orc_norm0, moving0, steps_checked, clip=False):
  world = len(results)
  for rank, (state, losses, norms) in enumerate(results):
    # every worker's losses on ITS batch, step by step (step k's loss depends on the update of step k - 1)
    for step in range(steps_checked):
      for k, exp in exp_losses[step][rank].items():
        got = losses[step][k]
        assert abs(got - exp) <= 1e-4 * max(1.0, abs(exp)), (rank, step, k, got, exp)
    if clip:
      assert abs(norms[0] - orc_norm0) <= 1e-4 * orc_norm0, (rank, norms[0], orc_norm0)
  if len(exp_losses) == 1:
    # one step: Adam's first moment = (1 - beta1) * mean gradient, for every variable, on every rank
    gmax = max(float(np.max(np.abs(v))) for k, v in orc_first.items() if k.endswith('/m'))
    for rank, (state, _, _) in enumerate(results):
      n_cmp = 0
      for key, ref in orc_first.items():
        name = key[:-2]
        if not key.endswith('/m') or key not in state:
          continue
        if name.endswith('/bias') and (name[:-len('/bias')] + '/bn/gamma') in orc.state:
          continue  # d(loss)/d(bias) == 0 under BatchNorm
        d, scale = float(np.max(np.abs(state[key] - ref))), float(np.max(np.abs(ref)))
        assert d <= 3e-4 * scale + 1e-6 * gmax, (rank, key, d, scale)
        n_cmp += 1
      assert n_cmp > 20
      # BatchNorm moving statistics are per worker (never synchronised)
      for k, ref in moving0[rank].items():
        assert np.allclose(state[k], ref, rtol=1e-4, atol=1e-6), (rank, k)
    if world > 1:
      a = next(k for k in moving0[0] if k.endswith('moving_mean'))
      assert not np.allclose(moving0[0][a], moving0[1][a]), 'the ranks were meant to see different batches'


def skew_to_owner0(cfg, batches, B, world):
  """Replace every batch's id strings by pre-hashed ids that all satisfy id % world == 0: one owner gets every key."""
  from oracle import hashing
  feats = [f for f in cfg.feature_config.features
           if f.feature_type == f.IdFeature and f.HasField('hash_bucket_size') and f.hash_bucket_size > 0]
  nb = np.array([f.hash_bucket_size for f in feats], dtype=np.uint64)
  for step in batches:
    for b in step:
      ids = hashing.hash_bucket_fast(np.asarray(b.pop('str_bytes')), np.asarray(b.pop('str_offsets')), B, nb,
                                     True).reshape(len(feats), B)
      b['hash_ids'] = np.where(ids >= 0, ids - (ids % world), ids).astype(np.int64)
