"""The headline step with the frames arriving from HOST memory ("PCIe-inclusive"): 256 x 1024^2 uint16 frames in
pinned host memory -> one asynchronous copy to the device -> EpidPipeline.run -> the [N, 9] records back to the host.  Never the
bench's `value` (that is measured with the frames resident in HBM); this is what a caller pays who hands over host buffers.
    python scripts/time_pcie_inclusive.py [frames=256] [passes=10]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pylinac_amd.pipeline import EpidPipeline  # noqa: E402
from pylinac_amd.synthetic import epid_open_field_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
host = epid_open_field_frames(n, 1024, 1024, seed0=1000, device=dev).cpu().pin_memory()
pipe = EpidPipeline(n, 1024, 1024, dev)
stage = torch.empty_like(host, device=dev)


def one():
    stage.copy_(host, non_blocking=True)
    return pipe.run(stage).record().cpu()


one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / passes
t1 = time.perf_counter()
for _ in range(passes):
    stage.copy_(host, non_blocking=True)
torch.cuda.synchronize()
dc = (time.perf_counter() - t1) / passes
print(f"pcie-inclusive: {dt * 1e3:.3f} ms per {n} frames = {n / dt:.0f} images/s; the copy alone {dc * 1e3:.3f} ms = "
      f"{host.numel() * 2 / dc / 1e9:.1f} GB/s host->device", flush=True)
