"""Driver for counter passes over the HEADLINE pipeline with as few framework kernels as possible (a --pmc pass serialises every
kernel of the process: the bench's 256-frame generator alone is ~5 000 launches): eight generated frames, repeated to `n`.
    python scripts/run_epid_pass.py [n=256] [passes=3]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd.pipeline import EpidPipeline  # noqa: E402
from pylinac_amd.synthetic import epid_open_field_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
base = epid_open_field_frames(8, 1024, 1024, seed0=1000, device=dev)
fr = base.view(torch.int16).repeat((n + 7) // 8, 1, 1)[:n].contiguous().view(torch.uint16)
pipe = EpidPipeline(n, 1024, 1024, dev)
pipe.run(fr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    pipe.run(fr)
torch.cuda.synchronize()
print(f"epid pass: {(time.perf_counter() - t0) / passes * 1e3:.3f} ms per {n} frames", flush=True)
