#!/usr/bin/env python
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup): calls, mean / min duration.
usage: trace_summary.py <kernel_trace.csv> [name filter substrings...]"""
import csv
import sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
filt = sys.argv[2:]
agg = defaultdict(list)
for r in rows:
  name = r['Kernel_Name']
  if filt and not any(f in name for f in filt):
    continue
  grid = (r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
  wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
  dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  agg[(name[:60], grid, wg, r.get('LDS_Block_Size', ''))].append(dur)
out = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
for (name, grid, wg, lds), d in out:
  print('%-60s grid %-18s wg %-5s lds %-6s calls %5d mean %8.2f min %8.2f us total %9.1f' %
        (name, 'x'.join(g for g in grid if g), wg, lds, len(d), sum(d) / len(d), min(d), sum(d)))
