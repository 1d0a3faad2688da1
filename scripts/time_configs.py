"""Throughput of the BASELINE.json parity configurations #3-#5 (they are parity-test cases, not bench lines; this is
for DESIGN.md section 7 only): frames resident in HBM, `iters` timed passes after one warm-up, wall clock with
torch.cuda.synchronize().  Appends one JSON line per configuration to the output file as soon as it is measured.

    python scripts/time_configs.py gpurun_out/configs.jsonl [iters] [scale]
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import ct, picketfence, winston_lutz  # noqa: E402

out_path = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu")
g.manual_seed(5)


def roll_batch(base, shifts):
    b16 = base.view(torch.int16) if base.dtype == torch.uint16 else base
    out = torch.stack([torch.roll(b16, (int(dy), int(dx)), dims=(0, 1)) for dy, dx in shifts])
    return out.view(torch.uint16) if base.dtype == torch.uint16 else out


def timed(name, n, fn, note):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    line = {"config": name, "frames": n, "ms_per_pass": round(dt * 1e3, 3), "frames_per_s": round(n / dt, 1), "note": note}
    with open(out_path, "a") as f:
        f.write(json.dumps(line) + "\n")
    print(json.dumps(line), flush=True)


# ---- config #4 (per GPU share): Winston-Lutz per-image analysis, 1024^2 uint16, SURVEY 8d recipe (seed 3000 + i)
from pylinac_amd.synthetic import wl_frames  # noqa: E402

n4 = max(int(512 * scale), 2)
frames4 = torch.from_numpy(wl_frames(n4)).to(dev)
timed("#4 WL analyze_batch (inversion check, clean edges, field CAX, BB sweep)", n4,
      lambda: winston_lutz.analyze_batch(frames4, 1 / 0.336, 5.0), "1024x1024 uint16")
timed("#4a WL field CAX only", n4, lambda: winston_lutz.field_centroids_batch(frames4), "1024x1024 uint16")
from pylinac_amd import features  # noqa: E402
timed("#4b WL BB sweep only", n4, lambda: features.bb_centroids_batch(frames4, 1 / 0.336, 5.0), "1024x1024 uint16")
del frames4

# ---- config #5: CatPhan phantom ROI per slice, 512^2 int16
hs = int(512 * min(scale, 1.0)) if scale < 1 else 512
n5 = max(int(400 * scale), 2)
mmpp = 0.5 * 512 / hs
yy, xx = torch.meshgrid(torch.arange(hs, device=dev), torch.arange(hs, device=dev), indexing="ij")
r = torch.hypot((yy - hs * 0.49).double(), (xx - hs * 0.51).double()) * mmpp
sl = torch.full((hs, hs), -1000.0, device=dev, dtype=torch.float64)
sl[r < 100] = 60.0
sl[(r < 100) & (((yy // 9) + (xx // 7)) % 2 == 0)] = 95.0
slices = roll_batch(sl.to(torch.int16), torch.randint(-hs // 14, hs // 14, (n5, 2), generator=g))
timed("#5 CatPhan phantom ROI (scharr, gaussian, float Otsu, clear_border, fill, label, regionprops)", n5,
      lambda: ct.phantom_roi_batch(slices, mmpp), f"{hs}x{hs} int16")
del slices

# ---- config #3: picket fence, AS1000 geometry 768 x 1024 uint16, 10 pickets, Millennium leaves
hp, wp, dpmm = 768, 1024, 1 / 0.390625
n3 = max(int(512 * scale), 2)
xs = torch.arange(wp, device=dev, dtype=torch.float64)
prof = torch.zeros(wp, device=dev, dtype=torch.float64)
for k in range(10):
    prof += torch.exp(-0.5 * ((xs - (180 + k * 15 * dpmm + (k % 3) * 0.37)) / 3.1) ** 2)
frame = (2000 + 50000 * prof)[None, :].expand(hp, wp)
frame = (frame + ((torch.arange(hp, device=dev)[:, None] * 3 + torch.arange(wp, device=dev)[None, :]) % 7)).round()
frame = frame.to(torch.int32).to(torch.int16).contiguous()
dx = torch.randint(-60, 60, (n3,), generator=g)
frames3 = torch.stack([torch.roll(frame, int(d), dims=1) for d in dx]).view(torch.uint16)
timed("#3 picket fence (column mean, picket peaks, 60 leaves x 10 pickets windows, FWXM positions)", n3,
      lambda: picketfence.analyze_batch(frames3, dpmm, num_pickets=10), f"{hp}x{wp} uint16")
