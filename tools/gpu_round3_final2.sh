#!/bin/bash
# the bench lines + rocprofv3 kernel stats of tools/gpu_round3_final.sh again (its gpurun_out exceeded the 64 MiB that are
# copied back: the kernel trace), keeping only the summaries
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03final; mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1; nproc >> $O/device.txt; lscpu | grep "Model name" >> $O/device.txt
run() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], '| steady', round(s.get('ms_per_step_mean',0),4), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))
"; }
run bench_default
run dcnv2_f32 --config configs/dcn_v2_criteo.config --steady_steps 256 --precondition 256 --cpu_seconds 3
run dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16 --steady_steps 256 --precondition 256 --cpu_seconds 3
run din10m --config configs/din_taobao_10m.config --steady_steps 128 --precondition 128 --cpu_seconds 3
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 128 --precondition 128 --cpu_seconds 2 --parity_steps 1
EASYREC_AMD_EXACT_DECAY=1 run exact_decay --no_cpu_baseline --steady_steps 256
run uniform --ids uniform --no_cpu_baseline --steady_steps 256
run ep1_rccl --force_ep --rccl --no_cpu_baseline --steady_steps 0
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python bench.py --steps 100 --warmup 10 --no_cpu_baseline --steady_steps 0 --precondition 128 > $O/prof.log 2>&1
find $O/prof -type f ! -name "*stats*" -delete
find $O/prof -type f | head; du -sh $O
