#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X:
"EPID images/s, filter->threshold->profile->peak on 1024^2 batch; % HBM roofline".

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (Gaussian(5) -> median(3) -> Otsu -> threshold -> column
profile -> FWXM peak) over one batch of 256 synthetic 1024x1024 uint16 frames PER GPU
(BASELINE.json configs[1]); frames are resident in HBM before the timed region.  N > 1 shards
independent frames (weak scaling, no data-path collective) and ends each step with the one
all-gather of the per-image scalar records.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy
ALG_BYTES_PER_FRAME = 4_194_304  # read u16 frame + write u16 frame


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--cpu-frames", type=int, default=96,
                    help="bounded CPU-baseline sample (frames through the oracle on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(frames_host, n_sample):
    """The oracle (numpy glue over the same scipy routines the reference calls) on host cores."""
    from oracle import pylinac_oracle as oracle

    n_sample = min(n_sample, frames_host.shape[0])
    oracle.epid_pipeline_frame(frames_host[0])  # warm-up (imports, page faults)
    t0 = time.perf_counter()
    for i in range(n_sample):
        oracle.epid_pipeline_frame(frames_host[i])
    dt = time.perf_counter() - t0
    return {
        "value": round(n_sample / dt, 3),
        "unit": "images/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_sample} of the same synthetic 1024x1024 uint16 frames through oracle/pylinac_oracle.py "
                  f"(scipy.ndimage gaussian+median, Otsu, threshold, np.mean, scipy.signal.find_peaks), "
                  f"single thread, {dt:.1f} s; host has {os.cpu_count()} cores",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # PL_BENCH_FORCE_DIST=1 exercises the RCCL init / all-gather / barrier path with a single rank too
    if world > 1 or os.environ.get("PL_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist  # noqa: F811

        dist.init_process_group(backend="nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from pylinac_amd import dist as pdist
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    n, h, w = args.frames, args.height, args.width
    # weak scaling: every rank owns `n` frames; global frame index = rank*n + i -> seed 1000 + index
    frames = epid_open_field_frames(n, h, w, seed0=1000 + rank * n, device=dev)
    pipe = EpidPipeline(n, h, w, dev)

    def step(events=None):
        res = pipe.run(frames, events)
        rec = res.record()
        return pdist.all_gather_records(rec, n * world) if dist is not None else rec

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    events = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-stage kernel time from the HIP events recorded INSIDE the timed region
    stage_ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in events.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    dom_s = stage_ms[dominant] / 1e3
    achieved = n * ALG_BYTES_PER_FRAME / dom_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dominant)
        except Exception:
            traffic = None

    if rank == 0:
        total_frames = n * world * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        value = total_frames / elapsed
        line = {
            "metric": "EPID images/s, filter->threshold->profile->peak on 1024^2 batch",
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 256 synthetic 1024x1024 uint16 EPID frames per GPU: "
                            "Gaussian(5)+median(3)+Otsu threshold on MI355X, then column-mean profile + FWXM peak",
                "frames_per_gpu": n, "height": h, "width": w,
                "parallelism": f"{world} x independent frame shards + 1 all-gather of [N,9] f64 records",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "note": "dominant kernel reads+writes one u16 frame (4 MiB/frame algorithmic; PMC traffic "
                        "matches); it is VALU-issue-bound, not HBM-bound: scipy-exact float64 accumulation "
                        "decided in packed float32 (~47-52 VALU instr/px at sigma=5; float64 kernels 61/px) "
                        "(DESIGN.md section 5)",
                "pipeline_frac": round(value / world * ALG_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 4),
                "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(frames[: args.cpu_frames].cpu().numpy(), args.cpu_frames)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
