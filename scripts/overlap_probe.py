"""Does running the Gaussian passes of chunk i+1 (FP64-issue-bound) beside the memory-bound stages of
chunk i pay on MI355X?  Times EpidPipeline.run for chunks in {1,2,4,8,16} and checks the records agree."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylinac_amd.pipeline import EpidPipeline
from pylinac_amd.synthetic import epid_open_field_frames

dev = torch.device("cuda", 0)
n = 256
frames = epid_open_field_frames(n, 1024, 1024, seed0=1000, device=dev)
ref = None
for chunks in (1, 2, 4, 8, 16, 1):
    pipe = EpidPipeline(n, 1024, 1024, dev, chunks=chunks)
    for _ in range(2):
        rec = pipe.run(frames).record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        rec = pipe.run(frames).record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = rec.clone()
    same = bool(torch.equal(ref, rec))
    print(f"chunks={chunks:2d}  {dt*1e3:7.3f} ms/step  {n/dt:9.0f} img/s  records_equal={same}", flush=True)
    del pipe
