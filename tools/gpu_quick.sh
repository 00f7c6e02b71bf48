#!/bin/bash
# quick GPU check: selected tests + bench lines.  usage: gpu_quick.sh TAG "pytest -k expr" [bench args...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-q}; KEXPR=${2:-}; shift 2
O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -12 $O/pytest.log
fi
timeout 600 python bench.py --no_cpu_baseline "$@" > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log; tail -3 $O/bench.log | cut -c1-1500
