"""Phase stopwatch of mask_regions_kernel (config #5's labelling kernel) on the bench's slices: needs a library built with
-DPL_SR_TIMING=1 (scripts/build_lib_variant.sh srt . -DPL_SR_TIMING=1; PYLINAC_HIP_LIB=build/variants/lib_srt.so).
    PYLINAC_HIP_LIB=build/variants/lib_srt.so python scripts/time_sr_phases.py [volumes=25]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import _lib, ct  # noqa: E402
from pylinac_amd.synthetic import catphan_volume  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device("cuda:0")
vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(nv)]).to(dev)
x = vols.reshape(-1, 512, 512)
lib = _lib.load()
buf = (C.c_ulonglong * 32)()
ct.phantom_roi_batch(x, 0.5)
torch.cuda.synchronize()
lib.pl_debug_sr_timing(buf)
reps = 3
for _ in range(reps):
    ct.phantom_roi_batch(x, 0.5)
torch.cuda.synchronize()
lib.pl_debug_sr_timing(buf)
t = np.array(list(buf), dtype=np.float64) / (reps * x.shape[0]) / 100.0          # s_memtime ticks at 100 MHz -> us per workgroup
names = {0: "plane build (read the float32 plane, ballots)", 4: "flag + paint passes (clear_border, fill_holes)", 5: "region table + ROI choice"}
sub = ("runs per row", "prefix", "run extraction", "first-above links", "pointer jumping", "unions + flatten")
for k, nm in names.items():
    print(f"   {nm:<50s} {t[k]:8.2f} us per workgroup")
for base, lab in ((8, "clear_border labelling"), (14, "fill_holes labelling"), (20, "final labelling")):
    print(f"   {lab:<50s} {t[base:base + 6].sum():8.2f} us per workgroup: " + ", ".join(f"{s} {v:.2f}" for s, v in zip(sub, t[base:base + 6])))
print(f"   total {t.sum():.2f} us per workgroup ({x.shape[0]} slices, {reps} passes)")
