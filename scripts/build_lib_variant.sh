#!/bin/bash
# Development aid: build a complete variant of the product library into build/variants/lib_<tag>.so for scripts/gpu_ab_libs.sh.
#   scripts/build_lib_variant.sh <tag> [git-ref] [extra hipcc flags...]
# git-ref "." (default) = the working tree; anything else takes csrc/ and include/ from that commit.
set -e
cd "$(dirname "$0")/.."
TAG=$1; REF=${2:-.}; shift; shift || true
W=build/variants/src_$TAG
rm -rf $W; mkdir -p $W/pylinac_amd $W/obj build/variants
if [ "$REF" = "." ]; then cp -r pylinac_amd/csrc $W/pylinac_amd/csrc; cp -r include $W/include
else git archive $REF pylinac_amd/csrc include | tar -x -C $W; fi
for f in $W/pylinac_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  [ $b = gaussian_mm ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $extra "$@" -c $f -o $W/obj/$b.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$TAG.so $W/obj/*.o
rm -rf $W
ls -la build/variants/lib_$TAG.so
