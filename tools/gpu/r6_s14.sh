#!/bin/bash
# round 6 session 14: step timelines (rocprofv3 --kernel-trace, rocpd): DeepFM with the sort and the lookup as separate
# launches (EASYREC_AMD_PROLOGUE_TABLES=0: which of the two the merged front launch waits for), DCN-v2 bf16, DIN, MMoE
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s14; mkdir -p $O
tl() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof_$name -o trace -- python bench.py --steps 100 --warmup 20 --no_cpu_baseline --steady_steps 0 --parity_steps 0 --precondition 64 "$@" > $O/$name.log 2>&1
  DB=$(find $O/prof_$name -name "*.db" | head -1)
  python tools/rocpd_timeline.py $DB 60 > $O/${name}_step_timeline.txt 2>&1; tail -3 $O/${name}_step_timeline.txt | cut -c1-200
  rm -rf $O/prof_$name; }
EASYREC_AMD_PROLOGUE_TABLES=0 tl default_sort_and_lookup_apart
tl dcnv2_bf16 --config configs/dcn_v2_criteo.config --dense_dtype bf16
tl dcnv2_f32 --config configs/dcn_v2_criteo.config
tl din10m --config configs/din_taobao_10m.config
tl mmoe25m --config configs/mmoe_taobao_4task_d64_25m.config
tl ep1_rccl --force_ep --rccl
ls $O; du -sh $O
