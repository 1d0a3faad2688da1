"""Development aid: the EPID step with the one-launch tail against the five-launch tail on one box (256 x 1024^2)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd.pipeline import EpidPipeline  # noqa: E402
from pylinac_amd.synthetic import epid_open_field_frames  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = 256, 1024, 1024
fr = epid_open_field_frames(n, h, w, seed0=1000, device=dev)
pipes = {True: EpidPipeline(n, h, w, dev, fused_tail=True), False: EpidPipeline(n, h, w, dev, fused_tail=False)}
ra, rb = pipes[True].run(fr), pipes[False].run(fr)
torch.cuda.synchronize()
same = all(torch.equal(getattr(ra, k).cpu(), getattr(rb, k).cpu()) for k in ("frames", "profile", "threshold", "status"))
print("identical results:", same, torch.equal(torch.nan_to_num(ra.fwxm.cpu(), nan=-1), torch.nan_to_num(rb.fwxm.cpu(), nan=-1)))
for rep in range(3):
    for fused in (False, True):
        p = pipes[fused]
        for _ in range(20):
            p.run(fr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            p.run(fr)
        torch.cuda.synchronize()
        print("fused_tail" if fused else "five launches", round((time.perf_counter() - t0) * 10, 4), "ms per step", flush=True)
