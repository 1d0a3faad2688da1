"""pl_edge_otsu alone, HIP events, for a sweep of batch sizes (development aid)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import ct, ops  # noqa: E402
from pylinac_amd.synthetic import catphan_volume  # noqa: E402

dev = torch.device("cuda", 0)
vol = catphan_volume(seed=4000, n_slices=80)
out = []
for n in (int(a) for a in (sys.argv[1:] or ["256", "512", "1024", "2000"])):
    x = torch.from_numpy(np.ascontiguousarray(np.concatenate([vol] * ((n + 79) // 80))[:n])).to(dev)
    spans = ct._disk_spans_on_device(512, 512, 0.5, dev)
    p32, rawmax, lo, hi = ops.edge_plane(x, 1, spans=spans)
    fn = lambda: ops.edge_otsu(p32, lo, hi, frames=x, sigma=1, spans=spans, scale=0.8)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    out.append(f"n={n}: {a.elapsed_time(b) / 10:.4f} ms")
    del x, p32
print("edge_otsu  " + "   ".join(out), flush=True)
