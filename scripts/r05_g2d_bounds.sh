#!/bin/bash
# Upper bounds on what is left in gauss2d_mm, from stopwatch builds of the kernel source that compute GARBAGE (scripts/ubench/
# gauss2d_variants.hip, -DPL_G2D_VARIANT=<bits>): 0 = the product kernel, 8 = no per-step barrier, 256 = a seven-MFMA tile
# (no level-1 pair, no undecided-pixel repair, minimal finish), 264 = both.  EPID-like frames, 256 x 1024^2, sigma 5.
# "sustained" = 2000 back-to-back launches; "window" = 25 launches after 300 ms of idleness (the bench's contract window).
cd $GRAFT_REPO_ROOT/scripts/ubench
for v in 0 8 256 264; do
  s=$(./g2d_v$v 256 1 2000 | head -1)
  w1=$(./g2d_v$v 256 1 25 | head -1)
  echo "variant $v  sustained(2000): $s"
  echo "variant $v  window(25):      $w1"
done
