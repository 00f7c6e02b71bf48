#!/bin/bash
# the DIN and MMoE bench lines (with full-size parity) on the round's last tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03ae; mkdir -p $O
run() { name=$1; shift; ( timeout 170 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}; p=d.get('parity_full_size') or {}; r=d.get('roofline') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], '| steady', round(s.get('ms_per_step_mean',0),4), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| launches', sum(f['launches_per_step'] for f in r.get('families', [])))
"; }
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --cpu_seconds 1 --parity_steps 1 --steps 50
run din10m --config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --cpu_seconds 1 --parity_steps 1 --steps 50
