#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
rocm-smi --showid 2>/dev/null | head -8 > $O/box.txt; hostname >> $O/box.txt
timeout 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k 'gemm_bf16_nt or lazy_dense' 2>&1 | tail -2 | tee $O/canary.txt
if ! grep -q passed $O/canary.txt || grep -q failed $O/canary.txt; then echo 'canary failed: bad box?'; exit 0; fi
run() { echo "--- $*" | tee -a $O/lines.log; env "$@" timeout 240 python bench.py --config configs/dcn_v2_criteo.config --no_cpu_baseline --steps 200 --steady_steps 256 --precondition 256 $EXTRA 2>&1 | grep '^{' | tail -1 | tee -a $O/bench_lines.jsonl | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}; r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'steady', round(s.get('ms_per_step_mean',0),4), d['dtype'], r.get('kernel'), round(r.get('achieved',0),1), r.get('unit'), 'frac', round(r.get('frac',0),3), 'loss', d.get('final_loss'))" | tee -a $O/lines.log; }
EXTRA="--dense_dtype f32" run A=1
EXTRA="--dense_dtype bf16" run EASYREC_AMD_BF16_NT=0
EXTRA="--dense_dtype bf16" run EASYREC_AMD_BF16_NT=1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --config $GRAFT_REPO_ROOT/configs/dcn_v2_criteo.config --dense_dtype bf16 --steps 500 --warmup 20 --no_cpu_baseline --steady_steps 0 --precondition 256 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats_dcn_v2_bf16.csv --steps 776 | tail -60 > $O/stats.txt
rm -rf $O/prof
