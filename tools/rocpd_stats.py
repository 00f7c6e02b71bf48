#!/usr/bin/env python
"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 `*_results.db` (rocpd sqlite
output of `rocprofv3 --kernel-trace --stats`), written as the CSV the older rocprofv3 emitted directly.
usage: rocpd_stats.py <results.db> [out.csv] [--steps N]   (--steps: also print launches and time per step)"""
import csv
import sqlite3
import sys


def kernel_stats(path):
  db = sqlite3.connect(path)
  cur = db.cursor()
  cols = [r[1] for r in cur.execute('pragma table_info(rocpd_kernel_dispatch)')]
  sym_cols = [r[1] for r in cur.execute('pragma table_info(rocpd_info_kernel_symbol)')]
  name_col = 'display_name' if 'display_name' in sym_cols else ('kernel_name' if 'kernel_name' in sym_cols else sym_cols[-1])
  q = ('select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) '
       'from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s '
       'order by 3 desc' % (name_col, name_col))
  assert 'start' in cols and 'end' in cols and 'kernel_id' in cols, cols
  return list(cur.execute(q))


def main():
  args = [a for a in sys.argv[1:] if not a.startswith('--')]
  steps = None
  if '--steps' in sys.argv:
    steps = int(sys.argv[sys.argv.index('--steps') + 1])
    args = [a for a in args if a != str(steps)]
  rows = kernel_stats(args[0])
  total = sum(r[2] for r in rows) or 1
  out = [('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')]
  for name, calls, tot, mn, mx in rows:
    out.append((name, calls, tot, round(tot / calls, 1), round(100.0 * tot / total, 3), mn, mx))
  if len(args) > 1:
    with open(args[1], 'w', newline='') as f:
      csv.writer(f).writerows(out)
  for r in out[:45]:
    print(','.join(str(x) for x in r)[:230])
  if steps:
    print('# per step: %.1f launches, %.1f us of kernel time' % (sum(r[1] for r in rows) / steps, total / steps / 1e3))


if __name__ == '__main__':
  main()
