"""Experiment: config #3's pass as K chunks staggered over two streams -- chunk k's memory-bound front (min / max, leaf profile,
peaks) runs while chunk k-1's vector-bound window kernel does.  The stagger point (an event in front of pl_pf_measure) is hooked in
from here; nothing in the product changes.
    python scripts/run_pf_pass_overlap.py [n=512] [passes=20] [chunks=2]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import _lib, picketfence  # noqa: E402
from pylinac_amd.synthetic import pf_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
fr = pf_frames(n, device=dev)
lib = _lib.load()
orig = lib.pl_pf_measure
state = {"event": None}


def hooked(*a):
    if state["event"] is not None:
        state["event"].record(torch.cuda.current_stream())
        state["event"] = None
    return orig(*a)


lib.pl_pf_measure = hooked
side = torch.cuda.Stream()
bounds = [(n * k // chunks, n * (k + 1) // chunks) for k in range(chunks)]


def plain():
    return [picketfence.analyze_batch(fr, 1 / 0.390625, num_pickets=10)]


def staggered():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    out, prev = [], None
    for k, (a, b) in enumerate(bounds):
        s = main if k % 2 == 0 else side
        ev = torch.cuda.Event()
        with torch.cuda.stream(s):
            if prev is not None:
                s.wait_event(prev)                    # this chunk's front starts when the previous chunk's front is done
            state["event"] = ev
            out.append(picketfence.analyze_batch(fr[a:b], 1 / 0.390625, num_pickets=10))
        prev = ev
    main.wait_stream(side)
    return out


ref = plain()[0]
got = staggered()
torch.cuda.synchronize()
pos = torch.cat([g.position for g in got])
same = torch.equal(torch.nan_to_num(pos, nan=-1.0), torch.nan_to_num(ref.position, nan=-1.0)) and torch.equal(
    torch.cat([g.status for g in got]), ref.status)
for name, fn in (("plain", plain), (f"staggered x{chunks}", staggered), ("plain", plain), (f"staggered x{chunks}", staggered)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / passes
    print(f"{name:14s}: {dt * 1e3:.3f} ms per {n} frames = {n / dt:.0f} frames/s   (same results: {same})", flush=True)
