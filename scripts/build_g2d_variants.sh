#!/bin/bash
# builds scripts/ubench/g2d_v<bits>[_<tag>] for the attribution variants given as arguments; G2D_EXTRA = extra -D flags,
# G2D_TAG = suffix naming them:   G2D_EXTRA=-DPL_G2D_SLOTS=5 G2D_TAG=s5 scripts/build_g2d_variants.sh 0 3
cd "$(dirname "$0")/.."
for v in ${@:-0 1 2 3}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result -DPL_G2D_VARIANT=$v \
        $G2D_EXTRA scripts/ubench/gauss2d_variants.hip -o scripts/ubench/g2d_v$v${G2D_TAG:+_$G2D_TAG} 2>/dev/null &
done
wait
