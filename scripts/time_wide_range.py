"""EPID pipeline step time on frames whose values span more than the one-pass Otsu window (38 912 grey levels): every
frame takes the gated two-kernel histogram path.  python scripts/time_wide_range.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pylinac_amd.pipeline import EpidPipeline
from pylinac_amd.synthetic import epid_open_field_frames

dev = torch.device("cuda:0")
fr = epid_open_field_frames(256, 1024, 1024, device=dev)
# the raw frames already span 0..65535 (noise tails); what matters is the range of the FILTERED frame: map the 1 % / 99 %
# quantiles of frame 0 (background level, field plateau) to 500 / 65000
q = torch.quantile(fr[0].to(torch.float32).flatten()[::16], torch.tensor([0.01, 0.99], device=dev))
lo, hi = float(q[0]), float(q[1])
wide = ((fr.to(torch.float32) - lo) * (64500.0 / (hi - lo)) + 500.0).round().clamp(0, 65535).to(torch.int32).to(torch.uint16)
for name, x in (("synthetic (plateau - background = %d)" % (hi - lo), fr), ("plateau - background stretched to 64500", wide)):
    pipe = EpidPipeline(256, 1024, 1024, dev)
    for _ in range(3):
        res = pipe.run(x)
    torch.cuda.synchronize()
    ev = {}
    t0 = time.perf_counter()
    for _ in range(10):
        res = pipe.run(x, ev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    st = {k: round(sum(a.elapsed_time(b) for a, b in v) / len(v), 4) for k, v in ev.items()}
    print(name, "ms/step %.4f" % dt, "flagged", int(pipe.flag.sum()), st)
