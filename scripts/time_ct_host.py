"""Development aid: where the HOST's time goes in one config #5 pass (ct.ctp528_batch over 25 resident volumes): wall-clock of the
host-side steps (each includes the wait for the device results it needs), against the pass.
    python scripts/time_ct_host.py [nv=25] [passes=8]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import ct  # noqa: E402
from pylinac_amd.synthetic import catphan_volume  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 25
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(nv)]).to(dev)
acc, marks = {}, []


def timed(name):
    fn = getattr(ct, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            acc[name] = acc.get(name, 0.0) + (t1 - t)
            marks.append((name, t, t1))
    setattr(ct, name, wrapper)


for name in ("_phantom_roi_launch", "ctp528_profiles_batch", "_ctp528_mtf_launch", "_phantom_roi_finish", "find_phantom_axes_batch",
             "_device_centres_disagree", "_ctp528_mtf_finish"):
    timed(name)
ct.ctp528_batch(vols, 0.5)
torch.cuda.synchronize()
acc.clear()
total = 0.0
for _ in range(passes):
    marks.clear()
    t0 = time.perf_counter()
    ct.ctp528_batch(vols, 0.5)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    total += time.perf_counter() - t0
print(f"pass {total / passes * 1e3:.3f} ms")
for k, v in acc.items():
    print(f"  {k:28s} {v / passes * 1e6:8.1f} us")
print("last pass, host timeline (us from the call):")
for name, a, b in marks:
    print(f"  {(a - t0) * 1e6:8.1f} .. {(b - t0) * 1e6:8.1f}  {name}")
print(f"  returned at {(t1 - t0) * 1e6:8.1f}")
