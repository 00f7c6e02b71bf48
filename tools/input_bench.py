#!/usr/bin/env python
"""Host-side input rates of the packed formats on the Criteo layout, next to the CSV reader's
(tools/csv_bench.py): CriteoInput over the three flat binary files of a part, ParquetInput over packed row groups.  What
is timed is everything the host does per batch up to the packed arrays the device buffers take (no GPU needed).
usage: python tools/input_bench.py [rows]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging  # noqa: E402

import numpy as np  # noqa: E402

logging.disable(logging.WARNING)
from easyrec_amd import kernels  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * B
rng = np.random.default_rng(0)
kernels.hip()
tmp = tempfile.mkdtemp()

# -- Criteo binary
from easyrec_amd.input.criteo_input import CriteoInput  # noqa: E402

cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
rng.integers(0, 2, size=n, dtype=np.int32).tofile(os.path.join(tmp, 'p0_label.bin'))
rng.random((n, 13), dtype=np.float32).tofile(os.path.join(tmp, 'p0_dense.bin'))
rng.integers(0, 2**32, size=(n, 26), dtype=np.uint32).tofile(os.path.join(tmp, 'p0_category.bin'))
try:
  best = 0.0
  for _ in range(3):
    inp = CriteoInput(cfg.data_config, list(cfg.feature_config.features),
                      {'label_path': [os.path.join(tmp, 'p0_label.bin')], 'dense_path': [os.path.join(tmp, 'p0_dense.bin')],
                       'category_path': [os.path.join(tmp, 'p0_category.bin')]}, batch_size=B)
    t0 = time.perf_counter()
    nb = sum(1 for _ in inp.batches())
    best = max(best, nb * B / (time.perf_counter() - t0))
  print('Criteo binary (memory-mapped parts -> packed batch): %.2f M examples/s (best of 3, %d batches)' % (best / 1e6, nb))
except Exception as e:  # noqa: BLE001
  print('Criteo binary: not run (%s)' % str(e)[:200])

# -- CSV with TagFeature / SequenceFeature columns (the Taobao layouts of BASELINE configs 4-5): the per-row Python split
#    against er_split_cells_host + one hash call per feature
from easyrec_amd.input.csv_input import CSVInput  # noqa: E402
from easyrec_amd.protos.feature_config_pb2 import FeatureConfig  # noqa: E402

for config in ('din_taobao_small.config', 'mmoe_taobao_small.config'):
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', config))
  feats = list(cfg.feature_config.features)
  sep = cfg.data_config.separator
  split = {f.input_names[0]: (f.separator or '|', 50 if f.feature_type == FeatureConfig.SequenceFeature else 4)
           for f in feats if f.feature_type in (FeatureConfig.TagFeature, FeatureConfig.SequenceFeature)}
  rows = 16 * 1024
  path = os.path.join(tmp, config + '.csv')
  with open(path, 'w') as f:
    for i in range(rows):
      cells = []
      for fld in cfg.data_config.input_fields:
        if fld.input_name in split:
          s, k = split[fld.input_name]
          cells.append(s.join('%d' % rng.integers(0, 100000) for _ in range(int(rng.integers(1, k + 1)))))
        else:
          cells.append('%d' % rng.integers(0, 2 if fld.input_name in cfg.data_config.label_fields else 1000))
      f.write(sep.join(cells) + '\n')
  for native in ('1', '0'):
    os.environ['EASYREC_AMD_NATIVE_SPLIT'] = native
    best = 0.0
    for _ in range(2):
      inp = CSVInput(cfg.data_config, feats, path, batch_size=1024, hash_on_host=True)
      inp.native_split = native == '1'
      t0 = time.perf_counter()
      nb = sum(1 for _ in inp.batches(num_epochs=1))
      best = max(best, nb * 1024 / (time.perf_counter() - t0))
    print('%s as CSV (%d tag / sequence columns), %s split: %.0f examples/s' %
          (config, len(split), 'native  ' if native == '1' else 'per-row', best))

# -- packed Parquet (the reference's embedding-parallel format): scalar id columns, one ragged id list, dense columns
try:
  import pyarrow as pa
  import pyarrow.parquet as pq
  from easyrec_amd.input.parquet_input import ParquetInput
  cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_parquet_small.config'))
  rows = 64 * 4096
  lens = rng.integers(0, 5, size=rows)
  offs = np.zeros(rows + 1, dtype=np.int32)
  offs[1:] = np.cumsum(lens)
  t1 = pa.ListArray.from_arrays(pa.array(offs), pa.array(rng.integers(0, 10**9, size=int(offs[-1])).astype(np.int64)))
  table = pa.table({'label': rng.integers(0, 2, size=rows).astype(np.int32), 's1': rng.integers(0, 10**12, size=rows).astype(np.int64),
                    's2': rng.integers(0, 1000, size=rows).astype(np.int64), 't1': t1,
                    'd1': rng.standard_normal(rows).astype(np.float32), 'd2': (rng.random(rows) * 100).astype(np.float32)})
  path = os.path.join(tmp, 'part-0.parquet')
  pq.write_table(table, path, row_group_size=4 * 4096)
  best = 0.0
  for _ in range(3):
    inp = ParquetInput(cfg.data_config, list(cfg.feature_config.features), path, batch_size=4096)
    t0 = time.perf_counter()
    nb = sum(1 for _ in inp.batches(num_epochs=1))
    best = max(best, nb * 4096 / (time.perf_counter() - t0))
  print('packed Parquet (deepfm_parquet_small layout, %d batches of 4096): %.2f M examples/s (best of 3)' % (nb, best / 1e6))
except Exception as e:  # noqa: BLE001
  print('packed Parquet: not run (%s)' % str(e)[:300])
