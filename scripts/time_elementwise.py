"""Development aid: streaming rate of the elementwise entry points on a resident batch (256 x 1024 x 1024 uint16 by default).
    python scripts/time_elementwise.py [n_frames]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
x = torch.randint(0, 60000, (n, 1024, 1024), dtype=torch.int32, device=dev).to(torch.uint16)
mn, mx = ops.minmax(x)
cases = {
    "minmax      (read 2 B)": (lambda: ops.minmax(x), 2),
    "ground      (2 B -> 2 B)": (lambda: ops.ground(x, mn=mn), 4),
    "normalize   (2 B -> 8 B)": (lambda: ops.normalize(x, mx), 10),
    "threshold   (2 B -> 2 B)": (lambda: ops.threshold(x, 30000), 4),
    "invert      (min/max + 2 B -> 2 B)": (lambda: ops.invert(x), 6),
}
for name, (fn, bytes_per_px) in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt * 1e3:.3f} ms per {n} frames = {x.numel() * bytes_per_px / dt / 1e12:.2f} TB/s", flush=True)
