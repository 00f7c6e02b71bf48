#!/usr/bin/env python
"""One step of a rocprofv3 kernel_trace.csv as a timeline: start offset, duration and idle gap before every kernel.
usage: trace_timeline.py <kernel_trace.csv> <marker kernel substring> [step index from the end, default 2]
The marker is a kernel that runs once at the start of a step (e.g. hyper_select)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = marks[-back - 1], marks[-back]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in rows[a:b]:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  gap = (s - prev_end) / 1e3
  busy += (e - max(s, prev_end)) / 1e3 if e > prev_end else 0
  print('%9.1f us  dur %7.1f  gap %7.1f  %s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'][:70],
                                                      r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
  prev_end = max(prev_end, e)
print('step span %.1f us, busy %.1f us, kernels %d' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, busy, b - a))
