#!/bin/bash
# the round's kept bench lines (profiles/r02_bench_lines.jsonl) + the default line with cpu_baseline and parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
timeout 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k 'gemm_bf16_nt or lazy_dense' 2>&1 | tail -2 | tee $O/canary.txt
if ! grep -q passed $O/canary.txt || grep -q failed $O/canary.txt; then echo 'canary failed: bad box?'; exit 0; fi
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
s=d.get('steady_state') or {}; r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}; p=d.get('parity_full_size') or {}; e=d.get('embedding_stage') or {}
print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'ex/s | steady', round(s.get('ms_per_step_mean',0),4), 'p99', round(s.get('ms_per_step_p99',0),4), 'catch_up', round(s.get('catch_up_ms_p50',0),4), 'flush', round(s.get('flush_decay_ms',0),2), '|', d['dtype'], r.get('kernel'), round(r.get('achieved',0),1), r.get('unit'), 'frac', round(r.get('frac',0),3), '| cpu', c.get('value'), c.get('cores'), '| parity', p.get('max_rel_loss_diff'), p.get('ok'), '| stage GBps', e.get('stage_GBps'), e.get('stage_frac_of_hbm_peak'))"; }
run() { name=$1; shift; echo "--- $name: $*" | tee -a $O/lines.log; ( time timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "^real|Error|Traceback" $O/$name.out | head -3; }
run default
run lazy_adam --optimizer lazy_adam --no_cpu_baseline --steady_steps 512
run uniform --ids uniform --no_cpu_baseline --steady_steps 512
run ep_w1 --force_ep --no_cpu_baseline
run ep_w1_rccl --force_ep --rccl --no_cpu_baseline
run dense_sweep --dense_sweep --no_cpu_baseline --steady_steps 0 --precondition 64 --steps 50
run dcnv2_f32 --config configs/dcn_v2_criteo.config --no_cpu_baseline --steady_steps 256 --precondition 256
run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 --no_cpu_baseline --steady_steps 256 --precondition 256
run din10m --config configs/din_taobao_10m.config --no_cpu_baseline --steady_steps 128 --precondition 128
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --no_cpu_baseline --steady_steps 128 --precondition 128
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 1024 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB $O/kernel_stats_default.csv --steps 2044 | tail -50 > $O/stats.txt; rm -rf $O/prof; tail -1 $O/stats.txt
timeout 300 python tools/trace_step_ops.py > $O/step_ops.txt 2>&1; grep -c "^lib\|^aten" $O/step_ops.txt; grep "^aten" $O/step_ops.txt | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
