#!/bin/bash
# Round 6, call aa: edge_stream32_kernel with 1 (lib_e32a1.so: up to r06zy), 2, 4 (product), 8 rows in flight per wave -- config #5's
# pass and the kernel's average, alternating on one box; the edge / CatPhan GPU tests on the product first.
TAG=${1:-r06aa}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "edge or ctp528 or catphan or volume or phantom or regions" -rf 2>&1 | tail -2 | tee -a $OUT/summary.txt
for round in 1 2; do
  for lib in build/variants/lib_e32a1.so build/variants/lib_e32a2.so "" build/variants/lib_e32a8.so; do
    export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
    echo "== library ${lib:-product (4 rows)}" | tee -a $OUT/summary.txt
    timeout 300 python scripts/run_ct_pass.py 25 8 | tee -a $OUT/summary.txt
  done
done
for lib in build/variants/lib_e32a1.so build/variants/lib_e32a2.so "" build/variants/lib_e32a8.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== kernel stats, library ${lib:-product (4 rows)}" | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | grep "ms per pass\|edge_stream32" | tee -a $OUT/summary.txt
done
