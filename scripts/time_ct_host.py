"""Where a pass of ct.ctp528_batch spends its wall time (25 volumes): each phase bracketed by torch.cuda.synchronize()."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import ct  # noqa: E402
from pylinac_amd.synthetic import catphan_volume  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device("cuda:0")
vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(nv)]).to(dev)
flat = vols.reshape(nv * 80, 512, 512)
ct.ctp528_batch(vols, 0.5)
torch.cuda.synchronize()


def lap(acc, name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


acc = {}
reps = 5
for _ in range(reps):
    t = time.perf_counter()
    roi = ct.phantom_roi_batch(flat, 0.5)
    t = lap(acc, "phantom_roi_batch (3 launches + 1 D2H)", t)
    fzx, fzy = ct.find_phantom_axes_batch(roi, nv)
    t = lap(acc, "find_phantom_axes_batch (host)", t)
    prof, idx = ct.ctp528_profiles_batch(flat, 0.5, fzx, fzy, slices_per_volume=80)
    t = lap(acc, "ctp528_profiles_batch", t)
    out = ct.ctp528_mtf_batch(prof)
    t = lap(acc, "ctp528_mtf_batch", t)
t0 = time.perf_counter()
for _ in range(reps):
    ct.ctp528_batch(vols, 0.5)
torch.cuda.synchronize()
print(f"whole pass: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
for k, v in acc.items():
    print(f"  {k:45s} {v / reps * 1e3:8.3f} ms")
