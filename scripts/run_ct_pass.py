"""Driver for profiling BASELINE config #5 at the bench's size: `passes` x ct.ctp528_batch over `nv` resident CatPhan volumes.
    python scripts/run_ct_pass.py [nv=25] [passes=3] [chunk_volumes]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd import ct  # noqa: E402
from pylinac_amd.synthetic import catphan_volume  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 25
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(nv)]).to(dev)
ct.ctp528_batch(vols, 0.5, chunk_volumes=chunk)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    ct.ctp528_batch(vols, 0.5, chunk_volumes=chunk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / passes
print(f"ct pass (chunk {chunk}): {dt * 1e3:.3f} ms per {nv * 80} slices = {nv * 80 / dt:.0f} slices/s", flush=True)
