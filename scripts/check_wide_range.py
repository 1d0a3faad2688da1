"""Development aid: Otsu of the 3x3 medians of WIDE-range 1024^2 frames (every frame takes pl_median3_otsu16's fallback) against the
separate entry points (median plane -> exact histogram -> Otsu scan) and, for two frames, against the oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import pylinac_oracle as o
from pylinac_amd import ops
from pylinac_amd.synthetic import epid_open_field_frames
from scipy import ndimage

dev = torch.device("cuda:0")
fr = epid_open_field_frames(8, 1024, 1024, device=dev)
q = torch.quantile(fr[0].to(torch.float32).flatten()[::16], torch.tensor([0.01, 0.99], device=dev))
wide = ((fr.to(torch.float32) - float(q[0])) * (64500.0 / float(q[1] - q[0])) + 500.0).round().clamp(0, 65535).to(torch.int32).to(torch.uint16)
g = ops.gaussian_filter(wide, 5)
thr, mn, mx, flag = ops.median3_otsu16(g)
med = ops.median_filter(g, 3)
thr2, mn2, mx2 = ops.otsu_from_hist(ops.histogram16(med), med.dtype)
print("flagged", int(flag.sum()), "fused == separate:", torch.equal(thr, thr2), torch.equal(mn, mn2), torch.equal(mx, mx2))
m = med[:2].cpu().numpy()
print("oracle:", [int(o.threshold_otsu(f)) for f in m], thr[:2].tolist())
