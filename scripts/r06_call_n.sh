#!/bin/bash
# Round 6, calls n, o: the ring kernel (n: whole box in LDS, four rows in flight per wave; o: compact annulus, radii in registers)
# two stopwatch builds (lib_ringv1.so: staging only, lib_ringv2.so: taps only) -- one box.
TAG=${1:-r06n}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "circle or ctp528 or catphan or volume" -rf 2>&1 | tail -3 | tee -a $OUT/summary.txt
for round in 1 2; do
  for ring in 1 0; do
    export PL_CIRCLE_RING=$ring
    echo "== PL_CIRCLE_RING=$ring" | tee -a $OUT/summary.txt
    for i in 1 2; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee -a $OUT/summary.txt
  done
done
export PL_CIRCLE_RING=1
for lib in "" build/variants/lib_ringv1.so build/variants/lib_ringv2.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== kernel stats, library ${lib:-product}" | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | head -9 | grep "ms per pass\|circle" | tee -a $OUT/summary.txt
done
