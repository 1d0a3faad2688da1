"""TEST / BENCH INFRASTRUCTURE ONLY -- never imported by the product (pylinac_amd/).

CPU baselines for bench.py's ``cpu_baseline`` objects: the oracle (numpy glue over the same scipy routines the reference
calls, pinned to the reference's goldens in tests/test_oracle_golden.py) timed on the host cores, single-thread and with a
``multiprocessing`` pool of one worker per core ("a multiprocessing.Pool(os.cpu_count()) number with
frames sharded per process").  Workers only touch numpy / scipy.
"""
from __future__ import annotations

import os
import time

import numpy as np


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------ per-unit work
def _epid(frames):
    from oracle import pylinac_oracle as o

    for f in frames:
        o.epid_pipeline_frame(f)
    return len(frames)


def _pf(frames, dpmm=1 / 0.390625):
    from oracle import pylinac_oracle as o

    for f in frames:
        o.pf_measure(o.normalize(o.ground(f)), dpmm, num_pickets=10)      # the constructor's ground()/normalize() + analyze()
    return len(frames)


def _wl(frames, dpmm=1 / 0.336):
    from oracle import pylinac_oracle as o

    for f in frames:
        o.wl_analyze_frame(f, dpmm, 5.0)
    return len(frames)


def _ct(volume, slices, mmpp=0.5):
    """per slice: phantom ROI (scharr -> gaussian -> Otsu -> clear_border -> fill -> label -> regionprops) + circle profile
    + relative MTF about the slice's own ROI centre"""
    from oracle import pylinac_oracle as o

    size = np.pi * 101 ** 2 / mmpp ** 2
    for s in slices:
        _, row = o.catphan_phantom_roi(volume[s], mmpp, size)           # row: area, bbox4, centroid (r, c), ...
        o.ctp528_slice(volume, int(s), (row[6], row[5]), mmpp)
    return len(slices)


_WORK = {"epid": _epid, "pf": _pf, "wl": _wl, "ct": _ct}


def _make_inputs(kind: str, seed: int, units: int):
    """inputs a worker generates for itself (numpy only; the GPU-generated EPID / PF frames are passed in instead)"""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pl_synthetic", os.path.join(root, "pylinac_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    if kind == "wl":
        return (syn.wl_frames(units, seed0=seed),)
    if kind == "ct":
        vol = syn.catphan_volume(seed, n_slices=7 + units, size=512, mm_per_pixel=0.5)
        return (vol, list(range(3, 3 + units)))
    raise ValueError(kind)


def run_task(task):
    """(kind, payload) -> (kind, units, t_start, t_end); payload = arrays, or ("gen", seed, units)"""
    kind, payload = task
    if isinstance(payload, tuple) and len(payload) == 3 and payload[0] == "gen":
        args = _make_inputs(kind, payload[1], payload[2])
    else:
        args = (payload,)
    t0 = time.time()
    units = _WORK[kind](*args)
    return kind, units, t0, time.time()


def single_thread(kind: str, *args):
    """-> (units per second, units, seconds) in the calling process"""
    t0 = time.perf_counter()
    units = _WORK[kind](*args)
    dt = time.perf_counter() - t0
    return units / dt, units, dt


def _warm(_):
    from oracle import pylinac_oracle  # noqa: F401  (numpy / scipy imports, page faults)
    return os.getpid()


def pool_throughput(tasks_by_kind: dict, cores: int | None = None, timeout_s: float = 240.0):
    """``tasks_by_kind`` = {kind: [task, ...]} (one task per worker).  A spawn-context pool of ``cores`` workers is
    warmed up (every worker imports the oracle), then each kind runs on its own: its wall clock goes from its first
    task's start (after input preparation) to its last task's end -> {kind: (units per second, units, wall seconds)}.
    Raises ``multiprocessing.TimeoutError`` after ``timeout_s`` per kind (a bench must never hang on its baseline)."""
    import multiprocessing as mp

    cores = cores or os.cpu_count() or 1
    ctx = mp.get_context("spawn")               # the parent holds a HIP context: never fork it
    out = {}
    with ctx.Pool(cores) as pool:
        pool.map_async(_warm, range(4 * cores), chunksize=1).get(timeout=timeout_s)
        for kind, tasks in tasks_by_kind.items():
            rs = pool.map_async(run_task, tasks, chunksize=1).get(timeout=timeout_s)
            units = sum(r[1] for r in rs)
            wall = max(r[3] for r in rs) - min(r[2] for r in rs)
            out[kind] = (units / wall, units, wall)
    return out
