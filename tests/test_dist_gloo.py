"""CPU: the N>1 path (frame sharding + the single all-gather of per-image records) with
world_size 2 on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pylinac_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = pdist.shard_range(n_total, rank, world)
        # the "record" of frame i is [i, 2i, ..., 9i]: stands in for the per-image scalars
        local = torch.arange(a, b, dtype=torch.float64)[:, None] * torch.arange(1, 10, dtype=torch.float64)[None, :]
        full = pdist.all_gather_records(local, n_total)
        # the pipelined form bench.py uses for N > 1: three "steps" whose gathers are in flight together, completed by wait()
        pending, outs = [], []
        for step in range(3):
            outs.append(pdist.all_gather_records(local + step, n_total, pending))
        for w in pending:
            w.wait()
        for step, o in enumerate(outs):
            want = torch.arange(n_total, dtype=torch.float64)[:, None] * torch.arange(1, 10, dtype=torch.float64)[None, :] + step
            assert torch.equal(o, want), (rank, step)
        assert len(pending) == (3 if n_total % world == 0 else 0)      # uneven shards take the blocking path
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_all_gather_records_world2(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.arange(n_total, dtype=np.float64)[:, None] * np.arange(1, 10, dtype=np.float64)[None, :]
    for _, full in got:
        assert np.array_equal(full, expect)


def test_all_gather_is_identity_without_process_group():
    x = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    assert pdist.all_gather_records(x) is x
