#!/bin/bash
# library kernel vs the bare core on the same shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
for shape in "8192 1152 256" "8192 256 1152" "4096 256 640"; do
  for v in 0 1 8 9 10 11; do timeout 60 tools/micro/gemm_core $v $shape 2>&1 | tee -a $O/gemm_core.log; done
  timeout 60 tools/micro/lib_gemm 1 $shape 0 2>&1 | tee -a $O/gemm_core.log
  timeout 60 tools/micro/lib_gemm 1 $shape 1 2>&1 | tee -a $O/gemm_core.log
  timeout 60 tools/micro/lib_gemm 0 $shape 0 2>&1 | tee -a $O/gemm_core.log
done
