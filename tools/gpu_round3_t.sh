#!/bin/bash
# f32 GEMM main loop: fragments read just in time, no scheduling fences (new) against the committed build (_ab/base)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
BASE=$GRAFT_REPO_ROOT/_ab/base/easyrec_amd/csrc
for shape in "8192 1152 256" "8192 256 1152" "4096 256 640" "204800 128 128" "4096 128 256"; do
  for lay in 0 1 2; do
    echo -n "new  " | tee -a $O/lib_gemm.log; timeout 60 tools/micro/lib_gemm $lay $shape 0 2>&1 | tee -a $O/lib_gemm.log
    echo -n "base " | tee -a $O/lib_gemm.log; LD_LIBRARY_PATH=$BASE timeout 60 tools/micro/lib_gemm $lay $shape 0 2>&1 | tee -a $O/lib_gemm.log
  done
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm or grouped or linear or batchnorm or bn" 2>&1 | tail -3 | tee $O/tests.log
line() { python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read())
except Exception as e:
  print('NO JSON', e); sys.exit(0)
r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step | gemm', ' '.join('%.1f' % f['us_per_step'] for f in r.get('families', []) if f['family']=='gemm'))
"; }
run() { name=$1; shift; echo "--- $name" | tee -a $O/lines.log; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | tee -a $O/bench_lines.jsonl | line | tee -a $O/lines.log; grep -E "Error|Traceback" $O/$name.out | head -3; }
D="--config configs/din_taobao_10m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
M="--config configs/mmoe_taobao_4task_d64_25m.config --steady_steps 64 --precondition 64 --no_cpu_baseline --parity_steps 0 --steps 50"
F="--no_cpu_baseline --steady_steps 128 --parity_steps 0"
C="--config configs/dcn_v2_criteo.config --steady_steps 128 --precondition 128 --no_cpu_baseline --parity_steps 0"
for which in new base; do
  if [ $which = base ]; then export EASYREC_AMD_LIB=$BASE/libeasyrec_hip.so; fi
  run mmoe_$which $M
  run din_$which $D
  run deepfm_$which $F
  run dcnv2_$which $C
done
