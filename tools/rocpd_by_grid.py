#!/usr/bin/env python
"""Kernel durations from a rocprofv3 rocpd database grouped by (kernel, grid): rocpd_by_grid.py <db> [name filter]."""
import sqlite3
import sys
from collections import OrderedDict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
sym_cols = [r[1] for r in cur.execute('pragma table_info(rocpd_info_kernel_symbol)')]
name_col = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
q = ('select s.%s, d.start, d.end, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch d join '
     'rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start' % name_col)
agg = OrderedDict()
prev = None
for name, st, en, gx, gy in cur.execute(q):
  if flt not in name:
    continue
  key = (name[:48], gx, gy)
  agg.setdefault(key, []).append(en - st)
for k, v in agg.items():
  v = sorted(v)
  print('%-48s grid (%d,%d) n=%4d  min %7.2f  med %7.2f  mean %7.2f us' % (k[0], k[1], k[2], len(v), v[0] / 1e3,
                                                                            v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
