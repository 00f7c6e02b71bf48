"""Which data seeds of tests/test_kv_embedding.py's MMoE case run free of ReLU ties between the HIP path and the CPU
oracle (strict tolerances: every row).  usage: python tools/scan_kv_mmoe_seeds_gpu.py seeds..."""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
logging.disable(logging.WARNING)
from tests import test_kv_embedding as t  # noqa: E402

for seed in [int(s) for s in sys.argv[1:]]:
  try:
    t._run('cuda:0', config='mmoe_kv_taobao_small.config', n_kv=4, row_tol=2e-3, tie_rows=0, data_seed=seed)
    print(seed, 'ok', flush=True)
  except AssertionError as e:
    print(seed, 'FAIL', str(e)[:120].replace('\n', ' '), flush=True)
