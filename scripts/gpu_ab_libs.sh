#!/bin/bash
# A/B of library builds on ONE box: for every build/variants/lib_*.so given, copy it over the product library and print the
# bench's stage times (no parity, no CPU baseline); the in-tree library is restored at the end
cd $GRAFT_REPO_ROOT
cp pylinac_amd/libpylinac_hip.so /tmp/lib_orig.so
for lib in "$@"; do
  cp $lib pylinac_amd/libpylinac_hip.so
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-configs --no-parity --steps ${STEPS:-30} --warmup ${WARM:-10} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']
print('%-40s step %.4f  gauss %.4f  otsu %.4f  thr %.4f  rest %.4f' % ('$(basename $lib)', d['ms_per_step'], s['gauss2d'], s['median3_otsu16'], s['median3_threshold_colsum'], s['colsum_to_mean']+s['find_peaks']+s['fwxm_record']))"
  done
done
cp /tmp/lib_orig.so pylinac_amd/libpylinac_hip.so
