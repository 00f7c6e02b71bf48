#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout 1200 python -m pytest "tests/test_kernels_gpu.py::test_lazy_dense_decay_equals_the_sweep" tests/test_deepfm_gpu.py tests/test_grad_clip.py tests/test_embedding_parallel_gpu.py tests/test_multi_rank_oracle_gpu.py tests/test_models_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 > $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E " $O/pytest.log | head -20
run() { echo "--- $*" | tee -a $O/ab.log; env "$@" timeout 600 python bench.py --no_cpu_baseline --steps 200 --steady_steps 1024 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}
print(round(d['ms_per_step'],4), 'steady', round(s.get('ms_per_step_mean',0),4), 'p99', round(s.get('ms_per_step_p99',0),4), 'catch_up', round(s.get('catch_up_ms_p50',0),4), 'flush', round(s.get('flush_decay_ms',0),2))" | tee -a $O/ab.log; }
run EASYREC_AMD_OVERLAP_FLUSH=0 EASYREC_AMD_FLUSH_WINDOWS=64
run EASYREC_AMD_OVERLAP_FLUSH=1 EASYREC_AMD_FLUSH_WINDOWS=64 EASYREC_AMD_FLUSH_BLOCKS=1024
run EASYREC_AMD_OVERLAP_FLUSH=0 EASYREC_AMD_FLUSH_WINDOWS=32
run EASYREC_AMD_OVERLAP_FLUSH=0 EASYREC_AMD_FLUSH_WINDOWS=128
run EASYREC_AMD_OVERLAP_FLUSH=0 EASYREC_AMD_FLUSH_WINDOWS=256
