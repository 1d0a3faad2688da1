#!/bin/bash
# Round 6, calls x2, x3: hist16_two_window_kernel with windows of 9 728 bins for pl_hist16_wl (x2: a -D build of the whole kernel against the product; x3: the product, which now picks them for Winston-Lutz frames only, against the earlier library lib_otsuold.so)
# epilogue under the other's main loop) against the product's 19 456 (152 KB, one per CU) -- config #4 clean and noisy, one box.
TAG=${1:-r06x2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
for lib in "" build/variants/lib_otsuold.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== library ${lib:-product}" | tee -a $OUT/summary.txt
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "hist or winston or wl or percentile or order" -rf 2>&1 | tail -2 | tee -a $OUT/summary.txt
done
for round in 1 2; do
  for lib in "" build/variants/lib_otsuold.so; do
    export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
    echo "== library ${lib:-product}" | tee -a $OUT/summary.txt
    timeout 300 python scripts/run_wl_pass.py 1250 8 | tee -a $OUT/summary.txt
    timeout 300 python scripts/run_wl_pass.py 1250 8 noise | tee -a $OUT/summary.txt
  done
done
for lib in "" build/variants/lib_otsuold.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== kernel stats, library ${lib:-product}" | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh wl wln 2>&1 | grep "ms per pass\|hist16" | tee -a $OUT/summary.txt
done
