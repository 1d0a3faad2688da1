"""Orientation timings of the "next"-row paths (none of them is the bench metric): one JSON line per path, appended to
the output file as soon as it is measured.  Inputs are generated on the device or on the host before the clock starts.

    python scripts/time_next_rows.py gpurun_out/next_rows.jsonl [iters] [scale]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pylinac_amd import canny, features, gamma, ops, planar, roi, xim  # noqa: E402

out_path = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)


def timed(name, units, unit_name, fn, note=""):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    line = {"path": name, "units": units, "unit": unit_name, "ms_per_pass": round(dt * 1e3, 3),
            "units_per_s": round(units / dt, 1), "note": note}
    with open(out_path, "a") as f:
        f.write(json.dumps(line) + "\n")
    print(json.dumps(line), flush=True)


s = lambda v, lo=8: max(int(v * scale), lo)  # noqa: E731

# f1: XIM decode, one 1280^2 int32 image
def xim_stream(pixels):
    """a compressed XIM pixel stream for `pixels` (what the decoder undoes): (W + 1) int32 values, then one 1 / 2 / 4-byte
    difference per pixel, sizes in a 2-bit-per-pixel lookup table"""
    h, w = pixels.shape
    flat = pixels.astype(np.int64).ravel()
    i = np.arange(w + 1, h * w)
    diffs = flat[i] - flat[i - 1] - flat[i - w] + flat[i - w - 1]
    codes = np.where(np.abs(diffs) < 128, 0, np.where(np.abs(diffs) < 32768, 1, 2)).astype(np.uint8)
    c4 = np.concatenate([codes, np.zeros((-len(codes)) % 4, np.uint8)]).reshape(-1, 4)
    lut = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    body = bytearray(flat[: w + 1].astype("<i4").tobytes())
    sizes = 1 << codes.astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    buf = np.zeros(int(offs[-1]), np.uint8)
    for code, dt in ((0, "<i1"), (1, "<i2"), (2, "<i4")):
        sel = np.flatnonzero(codes == code)
        raw = diffs[sel].astype(dt).view(np.uint8).reshape(len(sel), 1 << code)
        for b in range(1 << code):
            buf[offs[sel] + b] = raw[:, b]
    return lut, np.concatenate([np.frombuffer(bytes(body), np.uint8), buf])


side = s(1280, 16)
img = np.cumsum(rng.integers(-300, 300, (side, side)), axis=1).astype(np.int32)
lut, stream = xim_stream(img)
d_lut, d_stream = torch.from_numpy(lut).to(dev), torch.from_numpy(stream).to(dev)
timed("f1 XIM decode", 1, "image", lambda: xim.decode_xim_pixels(d_lut, d_stream, side, side, 4, device=dev), f"{side}x{side} int32")

# f2: canny + phantom outline on a 1024^2 float64 frame
n = s(1024, 48)
yy, xx = torch.meshgrid(torch.arange(n, device=dev), torch.arange(n, device=dev), indexing="ij")
frame = 0.2 + 0.5 * (((yy - n // 2).abs() < n // 5) & ((xx - n // 2).abs() < n // 5)).double()
frame = frame + torch.from_numpy(rng.normal(0, 0.01, (n, n))).to(dev)
timed("f2 canny (sigma 2, quantile thresholds)", 1, "frame", lambda: canny.canny(frame, sigma=2, low_threshold=0.001, high_threshold=0.01, use_quantiles=True), f"{n}x{n} float64")
timed("f2 canny_regions (canny + label + bbox table)", 1, "frame", lambda: planar.canny_regions(frame, sigma=2), f"{n}x{n} float64")

# f3: 64 disk ROIs on each of 80 slices
slices = torch.from_numpy(rng.integers(-1000, 1000, (s(80), 512, 512)).astype(np.int16)).to(dev)
centres = np.stack([256 + 120 * np.cos(np.linspace(0, 2 * np.pi, 64, endpoint=False)), 256 + 120 * np.sin(np.linspace(0, 2 * np.pi, 64, endpoint=False))], axis=1)
timed("f3 disk ROI statistics", slices.shape[0] * 64, "ROI", lambda: roi.disk_roi_stats_batch(slices, centres, 9.0), "512x512 int16, radius 9")

# f4: gamma_2d on a 512^2 pair, DTA 3 px
g = s(512, 24)
ref = torch.from_numpy(rng.uniform(0, 100, (g, g))).to(dev)
ev = ref * torch.from_numpy(rng.uniform(0.97, 1.03, (g, g))).to(dev)
timed("f4 gamma_2d (3 %, 3 px)", 1, "image pair", lambda: gamma.gamma_2d(ref, ev, dose_to_agreement=3, distance_to_agreement=3), f"{g}x{g} float64")

# a13: BB finder on 256 windows of 134^2
w = s(256)
wins = torch.from_numpy(rng.normal(0.5, 0.01, (w, 134, 134))).to(dev)
yy, xx = torch.meshgrid(torch.arange(134, device=dev), torch.arange(134, device=dev), indexing="ij")
wins = wins + 0.4 * (((yy - 66) ** 2 + (xx - 70) ** 2) < 55).double()[None]
timed("a13 BB finder (threshold sweep)", w, "window", lambda: features.find_features_batch(wins, 2.98, 2.5, 0.5), "134x134 float64")
