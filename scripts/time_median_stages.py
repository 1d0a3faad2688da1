"""Times the median-related stages of the EPID pipeline in isolation (256 x 1024 x 1024 uint16, inputs resident):
median plane written vs consumed on the fly.  python scripts/time_median_stages.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pylinac_amd import ops
from pylinac_amd.synthetic import epid_open_field_frames

dev = torch.device("cuda:0")
fr = epid_open_field_frames(256, 1024, 1024, device=dev)
g = ops.gaussian_filter(fr, 5)
med = ops.median_filter(g, 3)
thr, _, _ = ops.otsu16(med)
hist = torch.empty((256, 65536), dtype=torch.int32, device=dev)
out = torch.empty_like(g); out2 = torch.empty_like(g); cs = torch.empty((256, 1024), dtype=torch.int64, device=dev)


def t(name, fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / reps:.4f} ms")


t("median_filter (plane written)", lambda: ops.median_filter(g, 3, out=out) if False else ops.median_filter(g, 3))
t("otsu16(median plane)", lambda: ops.otsu16(med, hist=hist))
t("threshold_colsum(median plane)", lambda: ops.threshold_colsum_u16(med, thr, out=out, colsum=cs))
t("median3_otsu16 (on the fly)", lambda: ops.median3_otsu16(g, hist=hist, scratch=out2))
t("median3_threshold_colsum (on the fly)", lambda: ops.median3_threshold_colsum_u16(g, thr, out=out, colsum=cs))
