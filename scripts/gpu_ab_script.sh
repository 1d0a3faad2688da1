#!/bin/bash
# A/B of library builds on ONE box with an arbitrary driver: bash scripts/gpu_ab_script.sh "<command>" lib1.so lib2.so ...
# (each library is copied over the product library in turn; the in-tree one is restored at the end)
cd $GRAFT_REPO_ROOT
CMD=$1; shift
cp pylinac_amd/libpylinac_hip.so /tmp/lib_orig.so
for lib in "$@"; do
  cp $lib pylinac_amd/libpylinac_hip.so
  echo "== $(basename $lib)"
  for rep in 1 2; do $CMD 2>&1 | tail -${TAILN:-1}; done
done
cp /tmp/lib_orig.so pylinac_amd/libpylinac_hip.so
