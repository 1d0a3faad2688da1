#!/bin/bash
# Round 6, call w: the Otsu window kernel's scan with an odd number of bins per lane (no four-way LDS bank conflicts) and float64
# running class sums (no 64-bit integer multiply-add / conversions per bin) against the previous commit's library
# (build/variants/lib_otsuold.so) -- the headline's stage times and the 32-frame step, alternating on one box.
TAG=${1:-r06w}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "otsu or pipeline or epid or hist or threshold" -rf 2>&1 | tail -3 | tee -a $OUT/summary.txt
for round in 1 2 3; do
  for lib in "" build/variants/lib_otsuold.so; do
    export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
    for frames in 256 32; do
      timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --frames $frames 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-product}', 'frames', $frames, d['value'], d['ms_per_step'], d['roofline']['stage_ms'])" | tee -a $OUT/summary.txt
    done
  done
done
