"""TEST INFRASTRUCTURE ONLY: runs bench.py's main() on the emulated device so that the LAUNCH PATH of `bench.py --gpus N` under
torch.distributed.run (env rendezvous, per-rank sharding, the record all-gather with work handles in flight, the barrier +
max-over-ranks clock, rank 0's single JSON line) is exercised where there is no GPU.  The numbers it prints mean nothing.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tests/bench_dryrun.py --gpus 2 --steps 1 --warmup 1 --frames 2 --height 96 --width 128 ...

bench.py itself has no such switch (it refuses to run without a HIP device); everything that differs is patched in HERE:
the emulated kernel library (tests/emu_backend.py), the process-group backend (gloo for "nccl"), HIP events (a clock).
"""
import os
import sys
import time
from unittest import mock

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-6)


def main():
    from emu_backend import emulated_device

    import bench

    real_init = dist.init_process_group

    def init_gloo(backend=None, **kw):
        kw.pop("device_id", None)
        return real_init(backend="gloo", **kw)

    with emulated_device(), mock.patch.object(dist, "init_process_group", init_gloo), \
            mock.patch.object(torch.cuda, "Event", _Event), mock.patch.object(torch.cuda, "set_device", lambda d: None), \
            mock.patch.object(torch.cuda, "device_count", lambda: int(os.environ.get("WORLD_SIZE", "1"))):
        bench.main()


if __name__ == "__main__":
    main()
