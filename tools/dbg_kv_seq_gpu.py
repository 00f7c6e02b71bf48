#!/usr/bin/env python
"""Probe behind tests/test_kv_embedding.py::test_hash_table_sequence_features_match_the_oracle_on_the_gpu: per-step loss
differences between the GPU step and the oracle for MultiTowerDIN-small at B = 48, with the history sequences embedded
from hash-table tables and from dense tables, over a few seeds (is the step-2 difference the hash-table path's, or what
any two fp32 summation orders do to a 48-row BatchNorm model after two Adam steps?).  usage: python tools/dbg_kv_seq_gpu.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logging  # noqa: E402

logging.disable(logging.WARNING)
from easyrec_amd.input.synthetic import SyntheticBatches  # noqa: E402
from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402
from oracle.model_oracle import OracleTrainer  # noqa: E402

for kv in (True, False):
  for seed in (4, 5, 6, 7):
    for B in (48, 512):
      cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'din_taobao_small.config'))
      if kv:
        for f in cfg.feature_config.features:
          if f.feature_type == f.SequenceFeature:
            f.ev_params.max_capacity = 1 << 17
      est = EasyRecEstimator(cfg, device='cuda:0', batch_size=B, seed=seed).build()
      orc = OracleTrainer(cfg, est.state_dict(), batch_size=B)
      gen = SyntheticBatches(cfg.data_config, est.feature_configs, batch_size=B, seed=12 + seed)
      out = []
      for step in range(4):
        b = gen.next_batch()
        est.train_step(b)
        got, exp = est.loss_values(), orc.train_step(b)
        out.append(max(abs(got[k] - exp[k]) / max(1.0, abs(exp[k])) for k in exp))
      print('hash-table sequences' if kv else 'dense tables        ', 'seed', seed, 'B', B, ' '.join('%.1e' % d for d in out), flush=True)
