#!/bin/bash
# usage: tools/ab_export.sh <commit>   -> _ab/base = that commit's tree with its library built (ships with gpurun)
set -e
cd "$(dirname "$0")/.."
rm -rf _ab/base; mkdir -p _ab/base
git archive "$1" | tar -x -C _ab/base
make -C _ab/base/easyrec_amd/csrc -j8 > /dev/null
echo "base = $1" > _ab/base/AB_COMMIT
