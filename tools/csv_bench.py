#!/usr/bin/env python
"""Host-side input rate of CSVInput on the Criteo layout (40 tab-separated fields): native batch decode
(er_decode_csv_host + zero-copy string cells) against the line-by-line Python path.  No GPU needed.
usage: python tools/csv_bench.py [rows]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from easyrec_amd import kernels  # noqa: E402
from easyrec_amd.input.csv_input import CSVInput  # noqa: E402
from easyrec_amd.utils import config_util  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = config_util.get_configs_from_pipeline_file(os.path.join(ROOT, 'configs', 'deepfm_criteo_small.config'))
feats = list(cfg.feature_config.features)
B = 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * B
rng = np.random.default_rng(0)
with tempfile.NamedTemporaryFile('w', suffix='.tsv', delete=False) as f:
  for i in range(n):
    fl = ['%d' % rng.integers(0, 500) if rng.random() > 0.2 else '' for _ in range(13)]
    c = ['%08x' % rng.integers(0, 2**32) if rng.random() > 0.2 else '' for _ in range(26)]
    f.write('\t'.join(['%d' % (i % 2)] + fl + c) + '\n')
  path = f.name
kernels.hip()  # load the library once (host entry points only)
for native, threads in (('1', '0'), ('1', '1'), ('0', '1')):
  os.environ['EASYREC_AMD_NATIVE_CSV'] = native
  os.environ['EASYREC_AMD_CSV_THREADS'] = threads
  for host_hash in (False, True):
    best = 0.0
    for _ in range(3):
      inp = CSVInput(cfg.data_config, feats, path, batch_size=B, hash_on_host=host_hash)
      t0 = time.perf_counter()
      nb = sum(1 for _ in inp.batches())
      best = max(best, nb * B / (time.perf_counter() - t0))
    print('native decode %s (%s), ids hashed on the %s: %.0f examples/s (best of 3)' %
          ('on ' if native == '1' else 'off', ('%d host threads' % min(os.cpu_count() or 1, 16)) if threads == '0' and native == '1'
           else 'one thread', 'host  ' if host_hash else 'device', best))
# the decode call alone, per 4096-line batch
text = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
kinds = [1] + [2] * 13 + [0] * 26
be = kernels.hip()
for threads in (1, 2, 4, 8, 0):
  bufs = {}  # (the output arrays are reused, as CSVInput reuses them: fresh ones are page-faulted in on every call)
  be.decode_csv_host(text, '\t', kinds, B, threads=threads, out=bufs)
  t0 = time.perf_counter()
  reps = 50
  for _ in range(reps):
    be.decode_csv_host(text, '\t', kinds, B, threads=threads, out=bufs)
  dt = (time.perf_counter() - t0) / reps
  print('decode of one %d-line batch, threads=%d: %.2f ms = %.2f M lines/s' % (B, threads, dt * 1e3, B / dt / 1e6))
os.unlink(path)
