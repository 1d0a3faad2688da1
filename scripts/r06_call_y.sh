#!/bin/bash
# Round 6, call y: stopwatch builds of hist16_two_window_kernel (-DPL_TW_VARIANT=1 no tally, 2 no tile maxima, 3 no table zeroing,
# 4 no 1/16 sample; wrong results, timing only) on config #4's noise-free frames: where do its 0.82 ms per 1 250 frames go?
TAG=${1:-r06y2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
cat > /tmp/run_hist.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import numpy as np
from pylinac_amd import ops
from pylinac_amd.synthetic import wl_frames
dev = torch.device("cuda:0")
n = 1250
fr = torch.from_numpy(wl_frames(n)).to(dev)
cnt = 1024 * 1024
ranks = np.array([0, cnt - 1, 100, cnt // 2, cnt - 100, 5000, cnt - 5000], dtype=np.int64)
for tag, f in (("clean", fr),):
    for _ in range(2):
        ops.histogram16(f, tiles=True, edge_window=2, ranks=ranks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.histogram16(f, tiles=True, edge_window=2, ranks=ranks)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: pl_hist16_wl {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us per {n} frames", flush=True)
PY
for lib in "" build/variants/lib_twv1.so build/variants/lib_twv2.so build/variants/lib_twv3.so build/variants/lib_twv4.so ""; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== library ${lib:-product}" | tee -a $OUT/summary.txt
  timeout 300 python /tmp/run_hist.py 2>/dev/null | tee -a $OUT/summary.txt
done
