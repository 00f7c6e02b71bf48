"""profiles/r04_pmc_by_kernel.json (tools/gpu/r4_final.sh; round 3: r03_pmc_by_kernel.json, tools/gpu_round3_pmc.sh: rocprofv3 --pmc
passes over the eager DeepFM step) ->
profiles/pmc_traffic.json `by_kernel`: HBM-side bytes per launch of every kernel of the step, which bench.py attaches to
its `roofline.traffic`.  Correction as MI355X_MICROARCH.md prescribes and profiles/r02_pmc_embedding.md calibrated:
FETCH_SIZE / WRITE_SIZE are KiB; a streaming kernel's 128-byte read requests are counted as 64 (fetch x 2); the random
64-byte rows of the embedding kernels are one request each, counted right (fetch x 1)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r03_pmc_by_kernel.json')
res = json.load(open(src))['default']
RANDOM_ROWS = ('emb_fwd_kernel', 'emb_bwd_tile', 'emb_bwd_own', 'emb_catch_up', 'emb_bwd_fix', 'gather_rows', 'emb_flush', 'emb_owner_serve',
               'emb_front_fwd')  # (emb_bwd_own_wgrad: the row update's random records counted right, the contraction's
                                 #  streamed operands at half - a LOWER bound; its GEMM half alone: profiles/r05_s11_pmc_fused_tail.txt)
by = {}
for name, c in res.items():
  if 'er::' not in name or 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
    continue
  key = name.split(' grid=')[0].replace('er::', '').strip()  # (template arguments kept: NN / NT / TN are different kernels)
  n = c['launches']
  d = by.setdefault(key, {'launches': 0, 'fetch_kib': 0.0, 'write_kib': 0.0})
  d['launches'] += n
  d['fetch_kib'] += c['FETCH_SIZE'] * n
  d['write_kib'] += c['WRITE_SIZE'] * n
out = {}
for key, d in by.items():
  f = 1 if any(p in key for p in RANDOM_ROWS) else 2
  fetch, write = d['fetch_kib'] / d['launches'], d['write_kib'] / d['launches']
  out[key] = {'bytes_per_launch': (f * fetch + write) * 1024.0, 'fetch_factor': f, 'FETCH_SIZE_KiB': fetch,
              'WRITE_SIZE_KiB': write, 'launches_averaged': d['launches'],
              'source': '%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, eager step)' % os.path.relpath(src, ROOT)}
p = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
cur = json.load(open(p))
cur['by_kernel'] = out
json.dump(cur, open(p, 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['bytes_per_launch'])[:12]:
  print('%-40s %8.2f MB / launch (fetch x%d)' % (k, v['bytes_per_launch'] / 1e6, v['fetch_factor']))
