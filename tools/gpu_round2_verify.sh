#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ff; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
( time timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout 200 2>&1 | grep -v "^$" | cut -c1-250 | tail -25 ) > $O/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E |^real" $O/pytest.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 300 python bench.py ) > $O/bench_default.out 2>&1; grep '^{' $O/bench_default.out | tail -1 > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); s=d['steady_state']; print(round(d['ms_per_step'],4), round(d['value']), 'steady', round(s['ms_per_step_mean'],4), 'parity', d['parity_full_size']['max_rel_loss_diff'], 'cpu', round(d['cpu_baseline']['value']), 'roofline', round(d['roofline']['frac'],3))"; grep real $O/bench_default.out
