#!/bin/bash
# the whole -m gpu suite + smoke + the bench lines of DESIGN.md section 6 that this round's later changes touch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02verify; mkdir -p $O
timeout 60 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('canary', float(x.sum()))" 2>&1 | tail -1 | tee $O/canary0.txt
if ! grep -q 'canary 1048576' $O/canary0.txt; then echo 'bad box'; exit 0; fi
( time timeout 600 python -m pytest tests -m gpu -q --tb=short --timeout 200 2>&1 | grep -v "^$" | cut -c1-250 | tail -30 ) > $O/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E |^real" $O/pytest.log | head -24
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}; p=d.get('parity_full_size') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), 'p99', round(s.get('ms_per_step_p99',0),4), '|', d['dtype'], round(r.get('achieved',0),1), r.get('unit'), 'frac', round(r.get('frac',0),3), '| cpu', c.get('value'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 400 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
rm -f $O/bench_lines.jsonl
run default
run dcnv2_f32 --config configs/dcn_v2_criteo.config --no_cpu_baseline --steady_steps 256 --precondition 256
run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 --no_cpu_baseline --steady_steps 256 --precondition 256
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128
