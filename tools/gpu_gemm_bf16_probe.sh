#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02u; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "bf16" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E " | head -20
for dbg in 0 1; do
cd /tmp && ER_NT_DEBUG=$dbg timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof$dbg -o g -- python $GRAFT_REPO_ROOT/tools/gemm_bf16_bench.py nt_only > $GRAFT_REPO_ROOT/$O/bench$dbg.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof$dbg -name "*.db" | head -1)
echo "== ER_NT_DEBUG=$dbg" | tee -a $O/by_shape.txt
python tools/rocpd_by_grid.py $DB gemm_bf16_nt | tee -a $O/by_shape.txt
rm -rf $O/prof$dbg
done
