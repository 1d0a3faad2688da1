#!/bin/bash
# Round 6, call m: the CTP528 ring staged in LDS (pl_circle_profile_ring) against the gather kernel it forwards to with
# PL_CIRCLE_RING=0 -- same library, same box, alternating; then the new GPU tests (ring parity, batched Starshot).
TAG=${1:-r06m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "circle or ctp528 or catphan or starshot or volume" -rf 2>&1 | tail -3 | tee -a $OUT/summary.txt
for round in 1 2; do
  for ring in 1 0; do
    export PL_CIRCLE_RING=$ring
    echo "== PL_CIRCLE_RING=$ring" | tee -a $OUT/summary.txt
    for i in 1 2; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee -a $OUT/summary.txt
  done
done
for ring in 1 0; do
  export PL_CIRCLE_RING=$ring
  echo "== kernel stats, PL_CIRCLE_RING=$ring" | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | head -9 | tee -a $OUT/summary.txt
done
