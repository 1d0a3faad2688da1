#!/bin/bash
# Round 6, call k: row segments of the packed-float32 edge kernel: at least 4 (product) / 2 / 1 x 32 waves per CU.
TAG=${1:-r06k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
for lib in "" build/variants/lib_e32w2.so build/variants/lib_e32w1.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== library: ${lib:-product}" | tee -a $OUT/summary.txt
  for i in 1 2 3; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | head -4 | tee -a $OUT/summary.txt
done
