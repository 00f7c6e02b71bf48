#!/usr/bin/env python
"""One training step's kernel timeline from a rocprofv3 rocpd database: rocpd_timeline.py <db> [steps to average].

The dispatches are cut into steps at every occurrence of the step's first kernel (the most frequent period's head); per
position in the step: kernel, grid, mean duration, mean gap to the previous kernel's end - how much of a step is kernels and
how much is the launch boundary between dependent kernels."""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
n_avg = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sym_cols = [r[1] for r in cur.execute('pragma table_info(rocpd_info_kernel_symbol)')]
name_col = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
rows = list(cur.execute('select s.%s, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z from rocpd_kernel_dispatch d join '
                        'rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start' % name_col))
names = [r[0] for r in rows]
# the step's head: the kernel whose occurrences are spaced by the most common distance, taken from the tail of the run
tail = rows[-min(len(rows), 20000):]
tn = [r[0] for r in tail]
head = None
for cand, cnt in Counter(tn).most_common():
  if 'hyper_select' in cand or 'prologue' in cand:
    head = cand
    break
if head is None:
  head = Counter(tn).most_common()[-1][0]
idx = [i for i, n in enumerate(tn) if n == head]
steps = [tail[a:b] for a, b in zip(idx[:-1], idx[1:])]
L = Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == L][-n_avg:]
print('step head %s; %d launches per step; averaged over %d steps' % (head[:60], L, len(steps)))
tot_k = tot_g = 0.0
for pos in range(L):
  durs = [s[pos][2] - s[pos][1] for s in steps]
  gaps = [s[pos][1] - s[pos - 1][2] for s in steps] if pos else [0]
  d, g = sum(durs) / len(durs) / 1e3, sum(gaps) / len(gaps) / 1e3
  tot_k += d
  tot_g += g
  r = steps[0][pos]
  print('%2d %-64s grid %8d x %d  kernel %7.2f us  gap before %6.2f us' % (pos, r[0][:64], r[3], r[4], d, g))
span = [s[-1][2] - s[0][1] for s in steps]
print('kernels %.1f us + gaps %.1f us = %.1f us per step (first start to last end: %.1f us)' % (tot_k, tot_g, tot_k + tot_g, sum(span) / len(span) / 1e3))
