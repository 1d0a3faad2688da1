"""Development aid: where the HOST's time goes in one config #4 pass (winston_lutz.analyze_batch over 1 250 resident frames): wall
clock of the host-side steps (enqueue only, except the last which waits for the table), against the pass.
    python scripts/time_wl_host.py [n=1250] [passes=8]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import features, ops, winston_lutz  # noqa: E402
from pylinac_amd.synthetic import wl_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
fr = torch.from_numpy(wl_frames(n)).to(dev)
acc, marks = {}, []


def timed(mod, name):
    fn = getattr(mod, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            acc[name] = acc.get(name, 0.0) + (t1 - t)
            marks.append((name, t, t1))
    setattr(mod, name, wrapper)


for mod, name in ((ops, "_percentile_plan"), (ops, "histogram16"), (ops, "wl_decisions"), (ops, "field_cax"),
                  (features, "bb_centroids_batch"), (ops, "pack_columns")):
    timed(mod, name)
winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
torch.cuda.synchronize()
acc.clear()
total = 0.0
for _ in range(passes):
    marks.clear()
    t0 = time.perf_counter()
    winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
    t1 = time.perf_counter()
    total += t1 - t0
print(f"pass {total / passes * 1e3:.3f} ms")
for k, v in acc.items():
    print(f"  {k:24s} {v / passes * 1e6:8.1f} us")
print("last pass, host timeline (us from the call):")
for name, a, b in marks:
    print(f"  {(a - t0) * 1e6:8.1f} .. {(b - t0) * 1e6:8.1f}  {name}")
print(f"  returned at {(t1 - t0) * 1e6:8.1f}")
