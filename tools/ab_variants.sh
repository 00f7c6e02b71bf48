#!/bin/bash
# usage: tools/ab_variants.sh name1 "flags1" [name2 "flags2" ...]  ->  _ab/<name>/libeasyrec_hip.so built from the CURRENT
# sources with extra compiler flags (compile-time A/B knobs: -DER_TILE_PASSES_V1=4 ...); _ab/ is git-ignored and ships with
# gpurun; select a build with EASYREC_AMD_LIB=_ab/<name>/libeasyrec_hip.so (same-box comparisons)
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  rm -rf _ab/$name; mkdir -p _ab/$name/include _ab/$name/easyrec_amd
  cp -r easyrec_amd/csrc _ab/$name/easyrec_amd/csrc; cp include/*.h _ab/$name/include/
  rm -f _ab/$name/easyrec_amd/csrc/*.o _ab/$name/easyrec_amd/csrc/*.so
  make -C _ab/$name/easyrec_amd/csrc -j8 EXTRA_FLAGS="$flags" > /dev/null
  mv _ab/$name/easyrec_amd/csrc/libeasyrec_hip.so _ab/$name/libeasyrec_hip.so
  rm -rf _ab/$name/easyrec_amd _ab/$name/include
  echo "$name: $flags" > _ab/$name/FLAGS
  ls -la _ab/$name/libeasyrec_hip.so
done
