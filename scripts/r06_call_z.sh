#!/bin/bash
# Round 6, calls z2, z5: hist16_two_window_kernel A/B against a library without the change (z2: trips in a per-frame rotated order -- no gain, dropped; z5: per-stretch counts of the table pixels in LDS instead of reading the table back, against the previous commit lib_twprev.so)
# the rotation (lib_twnorot.so): the histogram launch alone and config #4's passes, clean and noisy, one box.
TAG=${1:-r06z2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "hist or winston or wl or percentile or order or tile" -rf 2>&1 | tail -2 | tee -a $OUT/summary.txt
sed -n '/^cat > \/tmp\/run_hist.py/,/^PY$/p' scripts/r06_call_y.sh | sed '1d;$d' > /tmp/run_hist.py
for round in 1 2; do
  for lib in "" build/variants/lib_twprev.so; do
    export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
    echo "== library ${lib:-product}" | tee -a $OUT/summary.txt
    timeout 300 python /tmp/run_hist.py 2>/dev/null | tee -a $OUT/summary.txt
    timeout 300 python scripts/run_wl_pass.py 1250 8 | tee -a $OUT/summary.txt
    timeout 300 python scripts/run_wl_pass.py 1250 8 noise | tee -a $OUT/summary.txt
  done
done
