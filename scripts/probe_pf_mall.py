"""Probe: does config #3's second read of the frames (the scaled column mean, after min / max) come out of the 256 MB memory-side
cache when both kernels walk the batch in pieces that fit it?   python scripts/probe_pf_mall.py"""
import sys

import torch

sys.path.insert(0, ".")
from pylinac_amd import _lib, ops  # noqa: E402
from pylinac_amd.synthetic import pf_frames  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = 512, 768, 1024
x = pf_frames(n, device=dev)
lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
prof = torch.empty((n, w), dtype=torch.float64, device=dev)


def run(piece):
    for lo in range(0, n, piece):
        xs = x[lo:lo + piece]
        vmin, vmax = ops.minmax(xs)
        gmax = vmax - vmin
        _lib.check(lib.pl_scaled_colmean(xs.data_ptr(), xs.shape[0], h, w, vmin.data_ptr(), gmax.data_ptr(),
                                         prof[lo:lo + piece].data_ptr(), st), "pl_scaled_colmean")


ref = None
for piece in (512, 256, 128, 64, 32, 512):
    run(piece)
    torch.cuda.synchronize()
    if ref is None:
        ref = prof.clone()
    assert torch.equal(ref, prof)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(piece)
    e1.record()
    torch.cuda.synchronize()
    print(f"pieces of {piece:4d} frames ({piece * h * w * 2 / 1e6:6.1f} MB): min/max + column mean {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us per {n} frames", flush=True)
