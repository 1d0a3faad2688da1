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


def _dry_run_args(gpus):
    return ["--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--frames", "2", "--height", "96", "--width", "128",
            "--no-cpu-baseline", "--no-configs", "--no-parity"]


def test_bench_plain_python_form_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the form the driver's recorded command line has)
    starts two ranks itself and prints ONE line with n_gpus = rccl_ranks = 2; a WORLD_SIZE that contradicts --gpus exits
    non-zero instead of timing one GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    script = os.path.join(root, "tests", "bench_dryrun.py")
    r = subprocess.run([sys.executable, script, *_dry_run_args(2)], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["frames_total"] == 4
    assert len(line["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in line["per_rank_ms_per_step"])
    bad = subprocess.run([sys.executable, script, *_dry_run_args(8)], capture_output=True, text=True, timeout=300, cwd=root,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "contradicts WORLD_SIZE" in bad.stderr and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_launch_path_two_ranks_dry_run():
    """`bench.py --gpus 2` the way the driver launches it (python -m torch.distributed.run --nproc-per-node 2, env
    rendezvous on 127.0.0.1), on the emulated device (tests/bench_dryrun.py): rank 0 prints ONE JSON line with n_gpus = 2,
    rccl_ranks = 2, weak scaling (frames_total = 2 x frames per GPU) and a positive value.  Proves the plumbing of the
    N > 1 bench, not its speed."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "bench_dryrun.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--frames", "2", "--height", "96", "--width", "128", "--no-cpu-baseline", "--no-configs",
           "--no-parity"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["frames_per_gpu"] == 2 and line["config"]["frames_total"] == 4
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in line
