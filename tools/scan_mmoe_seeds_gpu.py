"""Which estimator / data seeds of tests/test_models_gpu.py's small MMoE case are free of ReLU ties between the HIP path
and the CPU oracle (a tie = one example's gradient lands on the other side: ~1% of one column).  usage: seeds..."""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
logging.disable(logging.WARNING)
from tests import test_models_gpu as t  # noqa: E402

name = sys.argv[1]
for seed in [int(s) for s in sys.argv[2:]]:
  try:
    t._first_steps(t._cfg(name), 128, seed)
    print(name, seed, 'ok', flush=True)
  except AssertionError as e:
    print(name, seed, 'FAIL', str(e)[:160].replace('\n', ' '), flush=True)
