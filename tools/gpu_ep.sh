#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ep; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_parallel_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 600 python bench.py --no_cpu_baseline --force_ep --steps 50 > $O/bench_ep1.log 2>&1; tail -2 $O/bench_ep1.log | cut -c1-600
timeout 600 python bench.py --no_cpu_baseline --force_ep --steps 50 --optimizer lazy_adam > $O/bench_ep1_lazy.log 2>&1; tail -2 $O/bench_ep1_lazy.log | cut -c1-400
timeout 600 python bench.py --no_cpu_baseline --no_graph --steps 50 --optimizer lazy_adam > $O/bench_eager_lazy.log 2>&1; tail -2 $O/bench_eager_lazy.log | cut -c1-400
# nccl world=1 process group smoke (RCCL init + collectives on one rank)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force_ep --steps 20 --warmup 3 --no_cpu_baseline --optimizer lazy_adam > $O/bench_torchrun1.log 2>&1; tail -2 $O/bench_torchrun1.log | cut -c1-400
