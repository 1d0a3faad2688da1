"""CPU: the N > 1 path with REAL per-image records.  World size 2 on gloo; every rank runs the
analyzers on its own shard through the emulated kernels (tests/emu_backend.py) and the gathered records must equal the
single-process result bit for bit: EPID [N, 9], Winston-Lutz [N, 4], picket fence [N, leaves, pickets] and CatPhan
per-slice records split BY VOLUME."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    """small seeded inputs of the four analyzers (host numpy)"""
    sys.path.insert(0, ROOT)
    from pylinac_amd.synthetic import catphan_volume, epid_open_field_frames, pf_frames

    epid = epid_open_field_frames(5, 96, 128, seed0=11, field_mm=18.0).numpy()
    wl = np.load(os.path.join(ROOT, "tests", "golden", "wl.npz"))["frames"][[0, 1, 6]]
    pf = pf_frames(3, 200, 320, seed0=2000, pixel_mm=1.0, pickets=5, picket_spacing_mm=40.0, gap_mm=4.0, blur_mm=3.0).numpy()
    vols = np.stack([catphan_volume(4000 + v, n_slices=8, size=256, mm_per_pixel=0.98) for v in range(3)])
    return epid, wl, pf, vols


def _records(epid, wl, pf, vols, emulated):
    """the four local_fn closures over host arrays (tensors are created inside the emulated-device context)"""
    from pylinac_amd import ct, picketfence, winston_lutz
    from pylinac_amd.pipeline import EpidPipeline

    dev = torch.device("cuda:0")

    def f_epid(a, b):
        fr = torch.from_numpy(epid[a:b]).to(dev)
        return EpidPipeline(b - a, fr.shape[1], fr.shape[2], dev).run(fr).record() if b > a else torch.zeros((0, 9), dtype=torch.float64)

    def f_wl(a, b):
        if b == a:
            return torch.zeros((0, 4), dtype=torch.float64)
        return torch.from_numpy(winston_lutz.analyze_batch(torch.from_numpy(wl[a:b]).to(dev), 1 / 0.336, 5.0)["record"])

    def f_pf(a, b):
        if b == a:
            return torch.zeros((0, 1, 16), dtype=torch.float64)
        return picketfence.analyze_batch(torch.from_numpy(pf[a:b]).to(dev), 1.0, num_pickets=5).position

    def f_ct(a, b):
        if b == a:
            return torch.zeros((0, 10), dtype=torch.float64)
        r = ct.ctp528_batch(torch.from_numpy(vols[a:b]).to(dev), 0.98)
        return torch.from_numpy(np.concatenate([r["center"], r["rmtf"]], axis=1))

    return f_epid, f_wl, f_pf, f_ct


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_backend import emulated_device

    from pylinac_amd import dist as pdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        epid, wl, pf, vols = _inputs()
        with emulated_device() as emu:
            f_epid, f_wl, f_pf, f_ct = _records(epid, wl, pf, vols, emu)
            out = (pdist.sharded_records(len(epid), f_epid), pdist.sharded_records(len(wl), f_wl),
                   pdist.sharded_records(len(pf), f_pf), pdist.sharded_volume_records(len(vols), vols.shape[1], f_ct))
        q.put((rank, [t.numpy() for t in out]))
    finally:
        dist.destroy_process_group()


def test_sharded_records_world2_equal_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_backend import emulated_device

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    # single-process result while the ranks work
    epid, wl, pf, vols = _inputs()
    with emulated_device() as emu:
        f_epid, f_wl, f_pf, f_ct = _records(epid, wl, pf, vols, emu)
        want = [f_epid(0, len(epid)).numpy(), f_wl(0, len(wl)).numpy(), f_pf(0, len(pf)).numpy(), f_ct(0, len(vols)).numpy()]
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert want[0].shape == (5, 9) and want[1].shape == (3, 4) and want[3].shape == (24, 10)
    assert np.isfinite(want[3][:, 2]).sum() >= 6          # some slices do carry line pairs
    for _, outs in got:
        for name, a, b in zip(("epid", "wl", "pf", "catphan"), outs, want):
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), name
