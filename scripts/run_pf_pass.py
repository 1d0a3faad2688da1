"""Driver for profiling BASELINE config #3: `passes` x picketfence.analyze_batch over `n` resident 768 x 1024 frames.
    python scripts/run_pf_pass.py [n=512] [passes=3] [exact]      ("exact": every window evaluates numpy's float64 np.std sequence)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import picketfence  # noqa: E402
from pylinac_amd.synthetic import pf_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
fr = pf_frames(n, device=dev)
exact = len(sys.argv) > 3 and sys.argv[3] == "exact"
fn = lambda: picketfence.analyze_batch(fr, 1 / 0.390625, num_pickets=10, exact_deviation=exact)
fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / passes
print(f"pf pass: {dt * 1e3:.3f} ms per {n} frames = {n / dt:.0f} frames/s", flush=True)
