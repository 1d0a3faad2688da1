"""Driver for profiling BASELINE config #4: `passes` x winston_lutz.analyze_batch over `n` resident 1024 x 1024 frames.
    python scripts/run_wl_pass.py [n=1250] [passes=3] [notiles] [noise]     ("notiles": the field CAX reads every frame whole;
                                                                           "noise": RandomNoiseLayer(0.001) on every frame)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import winston_lutz  # noqa: E402
from pylinac_amd.synthetic import wl_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
fr = torch.from_numpy(wl_frames(n)).to(dev)
if "noise" in sys.argv[3:]:
    g = torch.Generator(device=dev)
    g.manual_seed(3000)
    for lo in range(0, n, 125):
        blk = fr[lo:lo + 125].to(torch.float32)
        blk += torch.randn(blk.shape, generator=g, device=dev) * (0.001 * 65535.0)
        fr.view(torch.int16)[lo:lo + 125] = blk.clamp_(0, 65535).to(torch.int32).bitwise_and_(0xFFFF).to(torch.int16)
tiles = "notiles" not in sys.argv[3:]
fn = lambda: winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0, tile_maxima=tiles)
fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / passes
print(f"wl pass: {dt * 1e3:.3f} ms per {n} frames = {n / dt:.0f} frames/s", flush=True)
