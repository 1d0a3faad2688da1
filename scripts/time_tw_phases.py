"""Development aid: phase stopwatch of hist16_two_window_kernel (a library built with -DPL_TW_TIMING=1: thread 0 of every workgroup
leaves wall-clock stamps in bins 65520.. of its frame's table) on config #4's frames.
    PYLINAC_HIP_LIB=build/variants/lib_twtime.so python scripts/time_tw_phases.py [n=1250] [noise]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import ops  # noqa: E402
from pylinac_amd.synthetic import wl_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
dev = torch.device("cuda:0")
fr = torch.from_numpy(wl_frames(n)).to(dev)
if "noise" in sys.argv[2:]:
    g = torch.Generator(device=dev)
    g.manual_seed(3000)
    for lo in range(0, n, 125):
        blk = fr[lo:lo + 125].to(torch.float32)
        blk += torch.randn(blk.shape, generator=g, device=dev) * (0.001 * 65535.0)
        fr.view(torch.int16)[lo:lo + 125] = blk.clamp_(0, 65535).to(torch.int32).bitwise_and_(0xFFFF).to(torch.int16)
cnt = 1024 * 1024
ranks = np.array([0, cnt - 1, 100, cnt // 2, cnt - 100, 5000, cnt - 5000], dtype=np.int64)
out = torch.zeros((n, 65536), dtype=torch.int32, device=dev)
for _ in range(3):
    ops.histogram16(fr, out=out, tiles=True, edge_window=2, ranks=ranks)
torch.cuda.synchronize()
st = out[:, 65520:65528].cpu().numpy().astype(np.int64) * 0.01          # us
names = ["LDS zero + edge strips", "1/16 sample", "table zeroing", "main loop", "ranks: counts of 64 bins", "ranks: block scan",
         "ranks: owners walk"]
d = np.diff(st, axis=1)
print(f"{n} frames{' with noise' if 'noise' in sys.argv[2:] else ''}: per workgroup, median (10th .. 90th percentile) us")
for k, name in enumerate(names):
    print(f"  {name:24s} {np.median(d[:, k]):7.1f}  ({np.percentile(d[:, k], 10):6.1f} .. {np.percentile(d[:, k], 90):6.1f})")
print(f"  {'whole workgroup':24s} {np.median(st[:, 7]):7.1f}")
