#!/bin/bash
# runs every built attribution variant of gauss2d_mm (scripts/ubench/g2d_v*) on the MI355X
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in $(ls scripts/ubench/g2d_v* | sort -V); do timeout 60 $b; timeout 60 $b 256 1; done 2>&1 | tee gpurun_out/g2d_variants.txt
