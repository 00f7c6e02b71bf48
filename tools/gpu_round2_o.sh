#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/tools/gemm_bf16_bench.py > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
python - $DB <<'PY' | tee $O/by_shape.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sym_cols = [r[1] for r in cur.execute('pragma table_info(rocpd_info_kernel_symbol)')]
name_col = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
cols = [r[1] for r in cur.execute('pragma table_info(rocpd_kernel_dispatch)')]
gx = 'grid_size_x' if 'grid_size_x' in cols else None
q = 'select s.%s, d.start, d.end%s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start' % (name_col, ', d.grid_size_x, d.grid_size_y' if gx else '')
from collections import defaultdict
agg = defaultdict(list)
for row in cur.execute(q):
  name = row[0]
  if 'gemm' not in name and 'cast' not in name: continue
  key = (name[:60],) + tuple(row[3:])
  agg[key].append(row[2] - row[1])
for k, v in agg.items():
  v = sorted(v)
  print('%-60s grid %s n=%4d  min %7.2f  med %7.2f  mean %7.2f us' % (k[0], k[1:], len(v), v[0] / 1e3, v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
PY
rm -rf $O/prof
