#!/usr/bin/env python
"""Per-kernel averages of the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_final.sh + the calibration factors
from tools/pmc_calibrate.py (known byte counts) -> JSON on stdout."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def per_kernel(d, counter):
  files = glob.glob(os.path.join(root, d, '*counter_collection.csv'))
  agg = collections.defaultdict(list)
  for f in files:
    for r in csv.DictReader(open(f)):
      if r['Counter_Name'] == counter:
        agg[r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]].append(float(r['Counter_Value']))
  return {k: {'launches': len(v), 'avg_KiB': sum(v) / len(v)} for k, v in agg.items()}


out = {'unit_note': 'rocprofv3 FETCH_SIZE / WRITE_SIZE are reported in KiB'}
cal_f, cal_w = per_kernel('cal_fetch', 'FETCH_SIZE'), per_kernel('cal_write', 'WRITE_SIZE')
copy_bytes = (1 << 28) * 4
sweep_bytes = 4000000 * 16 * 4 * 3
cal = {}
for name, known in (('er::stream_copy_kernel<4>', copy_bytes), ('er::adam_decay_sweep_vec4_kernel<4>', sweep_bytes)):
  for k in cal_f:
    if k.startswith(name):
      cal[name] = {'known_bytes_each_way': known,
                   'fetch_factor': known / (cal_f[k]['avg_KiB'] * 1024.0),
                   'write_factor': known / (cal_w[k]['avg_KiB'] * 1024.0) if k in cal_w else None}
out['calibration'] = cal
f, w = per_kernel('pmc_fetch', 'FETCH_SIZE'), per_kernel('pmc_write', 'WRITE_SIZE')
top = sorted(f, key=lambda k: -f[k]['avg_KiB'] * f[k]['launches'])[:10]
out['bench_kernels'] = {k: {'fetch_KiB': f[k]['avg_KiB'], 'write_KiB': w.get(k, {}).get('avg_KiB'),
                            'launches': f[k]['launches']} for k in top}
sw = [k for k in f if k.startswith('er::adam_decay_sweep_vec4_kernel')]
if sw:
  k = sw[0]
  fetch_b, write_b = f[k]['avg_KiB'] * 1024.0, w[k]['avg_KiB'] * 1024.0
  c = cal.get('er::adam_decay_sweep_vec4_kernel<4>') or cal.get('er::stream_copy_kernel<4>') or {}
  out['adam_decay_sweep_dim16'] = {
      'fetch_bytes_raw': fetch_b, 'write_bytes_raw': write_b,
      'guide_corrected_bytes_per_launch': 2.0 * fetch_b + write_b,  # gfx950: FETCH_SIZE counts 64 B per 128 B request
      'calibrated_bytes_per_launch': (fetch_b * c['fetch_factor'] + write_b * c['write_factor']) if c.get('write_factor') else None,
  }
print(json.dumps(out, indent=1))
