#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ep; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_parallel_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -12 $O/pytest.log
for opt in adam lazy_adam; do
  timeout 600 python bench.py --no_cpu_baseline --force_ep --steps 50 --optimizer $opt > $O/bench_ep1_$opt.log 2>&1; tail -1 $O/bench_ep1_$opt.log | cut -c1-330
  timeout 600 python bench.py --no_cpu_baseline --force_ep --no_graph --steps 50 --optimizer $opt > $O/bench_ep1_eager_$opt.log 2>&1; tail -1 $O/bench_ep1_eager_$opt.log | cut -c1-330
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force_ep --steps 20 --warmup 3 --no_cpu_baseline > $O/bench_torchrun1.log 2>&1; tail -1 $O/bench_torchrun1.log | cut -c1-330
