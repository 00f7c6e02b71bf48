#!/bin/bash
# usage: gpu_pytest.sh TAG <pytest args...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest -m gpu -q --timeout 900 "$@" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -30 $O/pytest.log
