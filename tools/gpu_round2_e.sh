#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout 600 python -m pytest "tests/test_kernels_gpu.py::test_batchnorm_fused_into_the_gemm_launch" "tests/test_deepfm_gpu.py::test_fused_batchnorm_gemms_change_no_bit" -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-300 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -20
for v in "EASYREC_AMD_FUSED_BN_GEMM=1" "EASYREC_AMD_FUSED_BN_GEMM=0" "EASYREC_AMD_FUSED_BN_GEMM=1" "EASYREC_AMD_FUSED_BN_GEMM=0"; do
  echo "== $v" >> $O/ab.log
  env $v timeout 600 python bench.py --no_cpu_baseline --steps 300 --steady_steps 0 --precondition 0 --ring 16 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))" >> $O/ab.log 2>&1
done
cat $O/ab.log
timeout 300 python tools/trace_step_ops.py > $O/step_ops.txt 2>&1; head -3 $O/step_ops.txt
