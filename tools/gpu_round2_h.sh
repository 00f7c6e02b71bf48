#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest "tests/test_kernels_gpu.py::test_lazy_dense_decay_equals_the_sweep" tests/test_deepfm_gpu.py tests/test_grad_clip.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -20
echo "--- fresh tables, ring 16" | tee -a $O/ab.log
bash tools/gpu_ab2.sh "--steps 300 --steady_steps 0 --precondition 0 --ring 16" 2 | tee -a $O/ab.log
echo "--- preconditioned 1024, ring 256, steady 1024" | tee -a $O/ab.log
bash tools/gpu_ab2.sh "--steps 200 --steady_steps 1024" 2 | tee -a $O/ab.log
for w in 64 128 512; do echo "--- head, EASYREC_AMD_FLUSH_WINDOWS=$w" | tee -a $O/ab.log; EASYREC_AMD_FLUSH_WINDOWS=$w timeout 600 python bench.py --no_cpu_baseline --steps 200 --steady_steps 1024 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'steady', round(s.get('ms_per_step_mean',0),4), 'catch_up', round(s.get('catch_up_ms_p50',0),4), 'flush', round(s.get('flush_decay_ms',0),2))" | tee -a $O/ab.log; done
