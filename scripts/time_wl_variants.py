"""Development aid: Winston-Lutz analyze_batch timing (clean and noisy frames) + result check against the first library variant."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from pylinac_amd import winston_lutz
from pylinac_amd.synthetic import wl_frames
dev = torch.device("cuda:0")
for name, kw, n in (("clean", {}, 512), ("noisy", dict(noise_sigma=0.001), 512)):
    fr = torch.from_numpy(wl_frames(n, **kw)).to(dev)
    fn = lambda: winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
    r = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    rec = r["record"] if isinstance(r, dict) and "record" in r else None
    chk = float(np.nansum(np.asarray(rec.cpu() if hasattr(rec, "cpu") else rec))) if rec is not None else None
    print(name, "ms per pass %.4f" % ((time.perf_counter() - t0) / 5 * 1e3), "status0", int((np.asarray(r["status"]) == 0).sum()), "checksum", chk, flush=True)
