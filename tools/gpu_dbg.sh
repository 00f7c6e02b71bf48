#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== eager"; timeout 300 python bench.py --config configs/dcn_v2_criteo.config --optimizer lazy_adam --no_cpu_baseline --no_graph --steps 10 --warmup 3 2>&1 | tail -3 | cut -c1-300
echo "== graph"; timeout 300 python bench.py --config configs/dcn_v2_criteo.config --optimizer lazy_adam --no_cpu_baseline --steps 10 --warmup 3 2>&1 | tail -12 | cut -c1-300
echo "== graph dcn v1"; timeout 300 python bench.py --config configs/dcn_criteo.config --optimizer lazy_adam --no_cpu_baseline --steps 10 --warmup 3 2>&1 | tail -2 | cut -c1-300
