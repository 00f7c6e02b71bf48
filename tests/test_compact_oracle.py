"""CPU: the oracle over COMPACT embedding tables (OracleTrainer(compact_ids=...): only the rows the run's batches look
up, fetched by id) gives exactly the losses and rows of the oracle over the full tables - the reduction that lets
bench.py's parity_full_size check BASELINE config 5 at its stated 200 M rows (51 GB of table, 153 GB with Adam's slots)
on a host.  TF-Adam's every-row decay acts on each row independently and a row no lookup reads influences nothing."""
import os

import numpy as np
import pytest

from easyrec_amd.utils import config_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('config,criteo', [('deepfm_criteo_small.config', True), ('mmoe_taobao_small.config', False),
                                           ('din_taobao_small.config', False)])
def test_compact_tables_give_the_full_tables_losses_and_rows(ref_backend, config, criteo):
  from easyrec_amd.input.criteo_synthetic import SyntheticCriteo
  from easyrec_amd.input.synthetic import SyntheticBatches
  from easyrec_amd.model.easy_rec_estimator import EasyRecEstimator
  from oracle.model_oracle import OracleTrainer
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
  B = 64
  est = EasyRecEstimator(cfg, device='cpu', batch_size=B, seed=3).build()
  gen = (SyntheticCriteo if criteo else SyntheticBatches)(cfg.data_config, est.feature_configs, batch_size=B, seed=5)
  warm = [gen.next_batch() for _ in range(2)]
  batches = [gen.next_batch() for _ in range(3)]
  full = OracleTrainer(cfg, est.state_dict(), batch_size=B)
  for b in warm:  # a trained state: nonzero Adam slots on rows the later batches may or may not touch
    full.train_step(b)
  state = {k: v.copy() for k, v in full.state.items()}
  slots = {k: v.copy() for k, v in full.slots.items()}
  step = full.global_step
  tables = {n: (t['rows'], t['dim']) for n, t in est.engine.tables.items()}
  probe = OracleTrainer(cfg, {k: v for k, v in state.items() if k not in tables}, batch_size=B)
  ids = probe.probe_ids(batches, tables)
  assert set(ids) == set(tables) and all(len(v) < tables[k][0] or tables[k][0] <= 64 for k, v in ids.items())
  cstate = {k: (v[ids[k]] if k in ids else v) for k, v in state.items()}
  cslots = {k: (v[ids[k.rsplit('/', 1)[0]]] if k.rsplit('/', 1)[0] in ids else v) for k, v in slots.items()}
  compact = OracleTrainer(cfg, cstate, batch_size=B, compact_ids=ids)
  compact.resume(step, cslots)
  ref = OracleTrainer(cfg, state, batch_size=B)
  ref.resume(step, slots)
  for b in batches:
    a, c = ref.train_step(b), compact.train_step(b)
    assert a == c, (a, c)
  for name, listed in ids.items():
    assert np.array_equal(ref.state[name][listed], compact.state[name]), name
    for s in ('/m', '/v'):
      if name + s in ref.slots:
        assert np.array_equal(ref.slots[name + s][listed], compact.slots[name + s]), name + s
