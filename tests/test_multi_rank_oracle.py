"""Embedding-parallel ranks with DIFFERENT batches per rank against the oracle's W-worker step (CPU: the host logic of
routing / exchange / 1/W scaling / per-rank BatchNorm through the oracle's stand-in backend; the HIP twin is
tests/test_multi_rank_oracle_gpu.py).  Also: the fixed-capacity exchange's overflow flag must stop the run."""
import numpy as np
import pytest

from _multi_rank import check_against_oracle, make_cfg, rank_batches, run_world, skew_to_owner0


@pytest.mark.parametrize('world,lazy,padded,clip,steps', [
    (2, False, True, 0.0, 1), (3, True, True, 0.0, 1), (2, True, False, 0.0, 1), (2, False, True, 0.05, 1),
    (2, True, True, 0.0, 3)])
def test_ranks_with_their_own_batches_match_the_w_worker_oracle(ref_backend, world, lazy, padded, clip, steps):
  cfg = make_cfg('deepfm_criteo_small.config', lazy=lazy, clip=clip)
  B = 24
  batches = rank_batches(cfg, list(cfg.feature_config.features), B, world, steps)
  out = run_world(cfg, 'cpu', B, world, batches, padded=padded)
  check_against_oracle(*out, steps_checked=steps, clip=clip > 0)


def test_unequal_owner_counts_and_overflow_flag(ref_backend):
  """Ids that all fall on ONE owner (id % W == 0): per-owner counts are as unequal as they get.  With room for them
  the step matches the oracle; with a capacity below the count the sticky overflow flag must raise where the losses
  are read back (the step's results are void)."""
  world, B = 2, 24
  cfg = make_cfg('deepfm_criteo_small.config', lazy=True)
  feats = list(cfg.feature_config.features)
  batches = rank_batches(cfg, feats, B, world, 1)
  skew_to_owner0(cfg, batches, B, world)
  out = run_world(cfg, 'cpu', B, world, batches, padded=True, recv_slack=2.0)
  check_against_oracle(*out, steps_checked=1)
  with pytest.raises(RuntimeError, match='routed more than'):
    run_world(cfg, 'cpu', B, world, batches, padded=True, recv_slack=1.0)
