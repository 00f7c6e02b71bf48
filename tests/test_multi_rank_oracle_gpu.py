"""-m gpu twin of tests/test_multi_rank_oracle.py: W embedding-parallel ranks as threads on ONE MI355X (gpurun exposes
one), every rank with its OWN batch, through the real HIP kernels, against the oracle's W-worker step
(per-worker BatchNorm, gradients averaged over the workers, row gradients summed at the owners / W)."""
import logging

import pytest

pytestmark = pytest.mark.gpu

from _multi_rank import check_against_oracle, make_cfg, rank_batches, run_world, skew_to_owner0  # noqa: E402

logging.disable(logging.WARNING)
DEV = 'cuda:0'


@pytest.mark.parametrize('world,lazy,padded,clip,steps', [
    (2, False, True, 0.0, 1), (4, True, True, 0.0, 1), (3, True, False, 0.0, 1), (2, False, True, 0.05, 1),
    (2, True, True, 0.05, 1), (2, True, True, 0.0, 3)])
def test_ranks_with_their_own_batches_match_the_w_worker_oracle(world, lazy, padded, clip, steps):
  cfg = make_cfg('deepfm_criteo_small.config', lazy=lazy, clip=clip)
  B = 128
  batches = rank_batches(cfg, list(cfg.feature_config.features), B, world, steps)
  out = run_world(cfg, DEV, B, world, batches, padded=padded)
  check_against_oracle(*out, steps_checked=steps, clip=clip > 0)


def test_unequal_owner_counts_and_overflow_flag():
  world, B = 2, 128
  cfg = make_cfg('deepfm_criteo_small.config', lazy=True)
  batches = rank_batches(cfg, list(cfg.feature_config.features), B, world, 1)
  skew_to_owner0(cfg, batches, B, world)
  out = run_world(cfg, DEV, B, world, batches, padded=True, recv_slack=2.0)
  check_against_oracle(*out, steps_checked=1)
  with pytest.raises(RuntimeError, match='routed more than'):
    run_world(cfg, DEV, B, world, batches, padded=True, recv_slack=1.0)
