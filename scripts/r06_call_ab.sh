#!/bin/bash
# Round 6, call ab: edge_stream32_kernel (four rows in flight) with at least 2 / 4 (product) / 8 x 32 waves per CU -- fewer, longer row
# segments re-read fewer halo rows; config #5's pass and the kernel's average on one box.
TAG=${1:-r06ab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
for round in 1 2; do
  for lib in build/variants/lib_e32w2.so "" build/variants/lib_e32w8.so; do
    export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
    echo "== library ${lib:-product (factor 4)}" | tee -a $OUT/summary.txt
    timeout 300 python scripts/run_ct_pass.py 25 8 | tee -a $OUT/summary.txt
  done
done
for lib in build/variants/lib_e32w2.so "" build/variants/lib_e32w8.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== kernel stats, library ${lib:-product (factor 4)}" | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | grep "ms per pass\|edge_stream32" | tee -a $OUT/summary.txt
done
rm -rf gpurun_out/prof_cfg
