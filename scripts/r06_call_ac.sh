#!/bin/bash
# Round 6, call ac: the fused median + Otsu window kernel with 2 (product) / 3 / 4 / 6 rows in flight per lane (PL_OTSU_AHEAD): one
# workgroup of 16 waves per CU has 32 KB on its way with two -- the headline's stage times at 256 and 32 frames, one box.
TAG=${1:-r06ac}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
for round in 1 2; do
  for lib in "" build/variants/lib_otsua3.so build/variants/lib_otsua4.so build/variants/lib_otsua6.so; do
    export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
    for frames in 256 32; do
      timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --frames $frames 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('${lib:-product (2)}', 'frames', $frames, d['value'], d['ms_per_step'], 'otsu', s['median3_otsu16'], 'parity', d.get('parity_sample', {}).get('ok'))" | tee -a $OUT/summary.txt
    done
  done
done
