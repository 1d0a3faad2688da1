"""Stage times of the CatPhan localisation (BASELINE config #5) on one GPU: HIP events around each entry point, 320 slices of
512 x 512 int16 (four synthetic volumes), 20 repetitions after 3 warm-ups.

    python scripts/time_ct_stages.py [n_slices]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import ct, ops  # noqa: E402
from pylinac_amd.synthetic import catphan_volume  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
dev = torch.device("cuda", 0)
vol = catphan_volume(seed=4000, n_slices=80)
x = torch.from_numpy(np.ascontiguousarray(np.concatenate([vol] * ((n + 79) // 80))[:n])).to(dev)


def timed(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:60s} {a.elapsed_time(b) / reps:8.4f} ms per {n} slices", flush=True)


for fn in ct.STAGE_TIMERS(x, 0.5):
    timed(*fn)
