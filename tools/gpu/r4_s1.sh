#!/bin/bash
# round 4, session 1: the new files->GPU tests (config 1 / f1 / f2), MMoE at its stated 200 M rows on one GPU next to the
# 25 M shard (same box), launch-floor microbenchmark, A/B of the graph-branch overlap and of kernarg placement
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s1; mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt 2>&1; nproc >> $O/device.txt; free -g | head -2 >> $O/device.txt
timeout 600 python -m pytest tests/test_files_to_gpu.py -q -m gpu -x --timeout 300 2>&1 | tail -15 | tee $O/tests_files.log
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_deepfm_gpu.py -q -m gpu -x -k "decay or lazy or closed" --timeout 300 2>&1 | tail -5 | tee $O/tests_decay.log
for k in 0 1; do HIP_FORCE_DEV_KERNARG=$k timeout 120 tools/micro/launch_floor 48 200 2>&1 | tee -a $O/launch_floor.txt; done
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'ms/step | steady', round(s.get('ms_per_step_mean',0),4), '|', ' '.join('%s %.1f/%s' % (f['family'][:8], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
Q="--no_cpu_baseline --parity_steps 0 --steady_steps 0 --steps 200 --warmup 20"
run deepfm_base $Q
EASYREC_AMD_OVERLAP_DENSE=1 run deepfm_overlap_dense $Q
HIP_FORCE_DEV_KERNARG=1 run deepfm_dev_kernarg $Q
HIP_FORCE_DEV_KERNARG=0 run deepfm_host_kernarg $Q
run deepfm_base_again $Q
M="--no_cpu_baseline --parity_steps 0 --steady_steps 64 --precondition 64 --steps 50"
run mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config $M
run mmoe200m --config configs/mmoe_taobao_4task_d64_200m.config $M
run mmoe200m_uniform --config configs/mmoe_taobao_4task_d64_200m.config --ids uniform $M
rocm-smi --showmeminfo vram 2>/dev/null | grep -i used | head -2 >> $O/device.txt
