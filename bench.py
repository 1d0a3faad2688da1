#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X:
"EPID images/s, filter->threshold->profile->peak on 1024^2 batch; % HBM roofline".

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: starts the N ranks itself through the line above; a WORLD_SIZE that
                                       contradicts --gpus is an error, never a silent one-GPU run)

A "step" is one pass of the hot path (Gaussian(5) -> median(3) -> Otsu -> threshold -> column
profile -> FWXM peak) over one batch of 256 synthetic 1024x1024 uint16 frames PER GPU
(BASELINE.json configs[1]); frames are resident in HBM before the timed region.  N > 1 shards
independent frames (weak scaling by default, `--scaling strong` splits the 256 frames over the ranks;
no data-path collective) and ends each step with the one all-gather of the per-image scalar records.
Rank 0 prints ONE JSON line.

At N = 1 the line also carries `configs`: BASELINE configs #3 (picket fence), #4 (Winston-Lutz) and #5
(CatPhan CTP528) on their SURVEY 8d seeded generators, each with its own fraction of the HBM-read
roofline and its own CPU baseline, and `cpu_baseline` (single thread + one worker per host core) for the
headline workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# torch is imported inside the functions: the CPU baseline's spawn-context worker processes re-import this module as
# __mp_main__ and must stay light (numpy / scipy only)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0          # measured float4 copy (same guide): the ceiling a streaming kernel can reach
ALG_BYTES_PER_FRAME = 4_194_304  # read u16 frame + write u16 frame
ALG_BYTES = {"#2": 4_194_304, "#3": 1_572_864, "#4": 2_097_152, "#5": 524_288,     # read-once roofline per frame / slice
             "f4": 1_600}                                                             # row f4: one 200-detector float64 profile


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step (weak) / in total (strong)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--cpu-frames", type=int, default=24,
                    help="bounded CPU-baseline sample (frames through the oracle on one core, per repeat: 3 warm-up + 5 timed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs #3-#5")
    ap.add_argument("--no-parity", action="store_true", help="skip the headline parity sample against the oracle")
    ap.add_argument("--sustained-steps", type=int, default=150,
                    help="N = 1: steps of the extra steady-state measurement taken AFTER the contract's timed region (0 = off)")
    return ap.parse_args()


def mfma_view(tiles, seconds):
    tops = tiles * 9 * 32768 / seconds / 1e12
    return {"tiles": tiles, "tops": round(tops, 1), "peak_tops_datasheet": 5000.0, "frac_of_datasheet": round(tops / 5000.0, 4),
            "peak_tops_measured": 3944.0, "frac_of_measured": round(tops / 3944.0, 4),
            "peaks": "datasheet = dense int8 spec; measured = MI355X_MICROARCH.md's v_mfma_i32_16x16x64_i8 rate (>= 3944 TOPS)"}


def timed_passes(fn, iters=8):      # (eight passes after one untimed: three sat inside the power governor's ramp, like the headline)
    import torch

    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, out


# ------------------------------------------------------------------------------------------------ parity samples
# After a configuration's timed passes a few of the units it just produced are compared with the CPU oracle's sequence
# (oracle/ = the checker, never the thing measured): `"parity_sample": {"units": k, "ok": true}` in the line.  Bars as
# in tests/test_gpu_bench_size.py: integer results / indices / picket positions exact, BB centroids 1e-9, profiles 1e-9.
def parity_epid(res, frames, k=8):
    import numpy as np

    from oracle import pylinac_oracle as o

    idx = np.linspace(0, frames.shape[0] - 1, k).astype(int)
    pick = lambda t: np.stack([t[int(i)].cpu().numpy() for i in idx])      # (torch has no indexed gather for uint16)
    ref_out, ref_prof, ref_rec = o.epid_pipeline(pick(frames))
    rec = pick(res.record())
    ok = (np.array_equal(pick(res.frames), ref_out) and np.array_equal(pick(res.profile), ref_prof)
          and np.array_equal(rec[:, :3], ref_rec[:, :3]) and np.allclose(rec, ref_rec, rtol=1e-12, atol=0, equal_nan=True))
    return {"units": int(k), "ok": bool(ok), "against": "oracle.epid_pipeline (scipy gaussian_filter + median_filter, "
            "Otsu, threshold, np.mean, scipy find_peaks): frames, profiles, records"}


def parity_pf(res, frames, dpmm, k=4):
    import numpy as np

    from oracle import pylinac_oracle as o

    ok = True
    for i in np.linspace(0, frames.shape[0] - 1, k).astype(int):
        raw = frames[i].cpu().numpy()
        ref = o.pf_measure(o.normalize(o.ground(raw)), dpmm, num_pickets=10)
        P = len(ref["peak_idxs"])
        pos = res.position[i, :, :P].cpu().numpy()
        ok &= int(res.picket_count[i]) == P and np.array_equal(res.picket_idx[i, :P].cpu().numpy(), ref["peak_idxs"])
        ok &= float(res.spacing[i]) == ref["spacing"] and np.array_equal(np.isnan(pos), np.isnan(ref["position"]))
        ok &= np.array_equal(pos[~np.isnan(pos)], ref["position"][~np.isnan(pos)])
    return {"units": int(k), "ok": bool(ok), "against": "oracle.pf_measure: picket indices, spacing, every leaf x picket position"}


def parity_wl(res, frames, dpmm, k=4):
    import numpy as np

    from oracle import pylinac_oracle as o

    ok = True
    for i in np.linspace(0, frames.shape[0] - 1, k).astype(int):
        fx, fy, bx, by, inv, crop = o.wl_analyze_frame(frames[i].cpu().numpy(), dpmm, 5.0)
        r = res["record"][i]
        ok &= (fx, fy) == (r[0], r[1]) and inv == bool(res["inverted"][i]) and crop == int(res["crop"][i])
        ok &= abs(bx - r[2]) < 1e-9 and abs(by - r[3]) < 1e-9 and int(res["status"][i]) == 0
    return {"units": int(k), "ok": bool(ok), "against": "oracle.wl_analyze_frame: field CAX exact, BB centroid 1e-9, inversion, crop"}


def parity_ct(res, vols, mmpp, k=8):
    import numpy as np

    from oracle import pylinac_oracle as o

    spv = vols.shape[1]
    v = vols.shape[0] - 1                                        # the last volume of the batch
    vol = vols[v].cpu().numpy()
    prof = res["profiles"]
    ok = True
    for s in np.linspace(3, spv - 4, k).astype(int):
        m = v * spv + int(s)
        p, rmtf = o.ctp528_slice(vol, int(s), tuple(res["center"][m]), mmpp)
        ok &= np.allclose(prof[m].cpu().numpy(), p, rtol=0, atol=1e-9)
        ok &= np.allclose(res["rmtf"][m], rmtf, rtol=1e-9, atol=1e-9, equal_nan=True)
    return {"units": int(k), "ok": bool(ok), "against": "oracle.ctp528_slice about the device's fitted centre: circle profile 1e-9, rMTF 1e-9"}


def bench_configs(dev, world=8):
    """BASELINE configs #3 / #4 / #5 on one GPU at the per-GPU share BASELINE states (512 frames; 10 000 / 8 = 1 250
    frames; 200 / 8 = 25 volumes), inputs resident in HBM, SURVEY 8d seeded generators; #4n = #4 with the reference's
    dark-current layer (RandomNoiseLayer(0.001)) on every frame."""
    import numpy as np
    import torch

    from pylinac_amd import ct, picketfence, winston_lutz
    from pylinac_amd.synthetic import catphan_volume, pf_frames, wl_frames

    out = {}

    def entry(key, workload, units, unit, dt, inputs, parity):
        rate = units / dt
        gbs = rate * ALG_BYTES[key[:2]] / 1e9
        out[key] = {"workload": workload, "units": units, "unit": unit, "ms_per_pass": round(dt * 1e3, 3),
                    "value": round(rate, 1), "algorithmic_GBs": round(gbs, 2),
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "frac_of_measured_copy": round(gbs / HBM_COPY_GBS, 5),
                    "inputs": inputs, "parity_sample": parity}

    # ---- "#2w": the headline step on frames stretched to the full 16-bit range (field plateau - background = 64 500 grey
    # levels: a 16-bit-normalised EPID).  Their FILTERED range exceeds the 38 912-bin LDS window of the one-pass Otsu kernel,
    # so every frame takes that stage's full-range kernel (65 536 packed 16-bit counters); the other stages are unchanged.
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    nw = 256
    fr = epid_open_field_frames(nw, 1024, 1024, seed0=1000, device=dev)
    q = torch.quantile(fr[0].to(torch.float32).flatten()[::16], torch.tensor([0.01, 0.99], device=dev))
    lo_q, hi_q = float(q[0]), float(q[1])
    wide = torch.empty_like(fr)
    for a in range(0, nw, 32):                            # (in blocks: the float32 detour of 256 frames would be 1 GiB more)
        blk = ((fr[a:a + 32].to(torch.float32) - lo_q) * (64500.0 / (hi_q - lo_q)) + 500.0).round().clamp(0, 65535)
        wide.view(torch.int16)[a:a + 32] = blk.to(torch.int32).bitwise_and_(0xFFFF).to(torch.int16)
    del fr, blk
    pipe_w = EpidPipeline(nw, 1024, 1024, dev)
    ev = {}
    dt, rw = timed_passes(lambda: pipe_w.run(wide, ev), iters=20)
    entry("#2w", "headline step (configs[1]) on 256 x 1024^2 uint16 frames stretched to the full 16-bit range: the median + Otsu "
                 "stage runs its full-range kernel for every frame", nw, "images/s", dt,
          "synthetic.epid_open_field_frames seed 1000+i, (x - q01) * 64500 / (q99 - q01) + 500, rounded, clipped",
          parity_epid(rw, wide))
    out["#2w"]["flagged_frames"] = int(pipe_w.flag.sum())
    out["#2w"]["stage_ms"] = {k: round(sum(a.elapsed_time(b) for a, b in v) / len(v), 4) for k, v in ev.items()}
    del wide, rw, pipe_w, ev
    torch.cuda.empty_cache()

    n3 = 512
    f3 = pf_frames(n3, device=dev)
    dt, r3 = timed_passes(lambda: picketfence.analyze_batch(f3, 1 / 0.390625, num_pickets=10))
    entry("#3", "PicketFence: 512 x (768 x 1024) uint16, column mean -> picket peaks -> 60-leaf x 10-picket windows -> "
                "FWXM positions", n3, "frames/s", dt, "synthetic.pf_frames seed 2000+i", parity_pf(r3, f3, 1 / 0.390625))
    del f3, r3
    n4 = 10000 // world
    host4 = wl_frames(n4)
    f4 = torch.from_numpy(host4).to(dev)
    dt, r4 = timed_passes(lambda: winston_lutz.analyze_batch(f4, 1 / 0.336, 5.0))
    wl_text = (f"Winston-Lutz: {n4} (= 10 000 / {world}, one GPU's shard) x 1024^2 uint16, inversion check -> clean edges -> "
               "field CAX (percentile threshold, fill holes, centre of mass) -> BB threshold sweep + weighted centroid")
    entry("#4", wl_text, n4, "frames/s", dt, "synthetic.wl_frames seed 3000+i (generate_winstonlutz recipe, noise-free)",
          parity_wl(r4, f4, 1 / 0.336))
    del r4
    # the same frames under the reference's dark-current layer: N(0, 0.001 * 65535) per pixel through clip_add
    # (RandomNoiseLayer, pylinac/core/image_generator/layers.py:396-407), seeded, generated on the device
    g = torch.Generator(device=dev)
    g.manual_seed(3000)
    f4n = torch.empty_like(f4)
    for lo in range(0, n4, 125):
        blk = f4[lo:lo + 125].to(torch.float32)
        blk += torch.randn(blk.shape, generator=g, device=dev) * (0.001 * 65535.0)
        # clip_add: clip to the dtype's range, then numpy's truncating cast (same bits through int16: torch copies uint16 that way)
        f4n.view(torch.int16)[lo:lo + 125] = blk.clamp_(0, 65535).to(torch.int32).bitwise_and_(0xFFFF).to(torch.int16)
    del f4, blk
    dt, r4n = timed_passes(lambda: winston_lutz.analyze_batch(f4n, 1 / 0.336, 5.0))
    entry("#4n", wl_text + "; every frame with RandomNoiseLayer(sigma = 0.001) dark-current noise", n4, "frames/s", dt,
          "synthetic.wl_frames seed 3000+i + N(0, 65.5) per pixel (torch generator seed 3000 on the device), clipped",
          parity_wl(r4n, f4n, 1 / 0.336))
    del f4n, r4n
    nv = 200 // world
    vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(nv)]).to(dev)   # [V, 80, 512, 512]
    dt, r5 = timed_passes(lambda: ct.ctp528_batch(vols, 0.5))
    entry("#5", f"CatPhan-504: {nv} volumes (= 200 / {world}, one GPU's shard) x 80 x 512^2 int16, per slice: phantom "
                "ROI (scharr, gaussian, Otsu, clear_border, fill, label, regionprops) -> axis fits -> +-3-slice max -> "
                "collapsed circle profile -> 8-region peak/valley rMTF", nv * 80, "slices/s", dt,
          "synthetic.catphan_volume seed 4000+v", parity_ct(r5, vols, 0.5))
    try:                                                    # (an extra row must not take the bench line down)
        hill_entry(dev, entry, out)
    except Exception as exc:
        out["f4h"] = {"error": repr(exc)}
    return out, vols


def hill_entry(dev, entry, out, n=4096):
    """SURVEY row f4 (not a BASELINE configuration): SingleProfile(edge_detection_method=INFLECTION_HILL) + inflection_data() for
    4 096 open-field detector profiles in one batch -- resampling, grounding, beam-centre normalisation (= the edge search and
    both Hill fits twice), no per-profile host call; compute-bound (MINPACK's Levenberg-Marquardt per penumbra), so `frac` is
    only there for uniformity.  Parity sample: the oracle's restated SingleProfile (scipy curve_fit) at the row's 1e-5."""
    import numpy as np
    import torch

    from oracle import pylinac_oracle as o
    from pylinac_amd import profile

    length = 200
    rng = np.random.default_rng(4100)
    x = np.arange(length, dtype=float) + 1.0
    profs = np.empty((n, length))
    for i in range(n):
        left, right, steep = rng.uniform(0.2, 0.3) * length, rng.uniform(0.7, 0.8) * length, rng.uniform(15, 40)
        dome = 1.0 - rng.uniform(0, 0.05) * ((x - (left + right) / 2) / length) ** 2
        profs[i] = (rng.uniform(50, 200) / (1.0 + (left / x) ** steep) / (1.0 + (x / right) ** (steep * right / left)) * dome
                    + rng.uniform(0, 2) + rng.normal(0, 0.05, length))
    d = torch.from_numpy(profs).to(dev)
    dt, res = timed_passes(lambda: profile.single_profile_hill_batch(d), iters=5)
    info = res.info.cpu().numpy()
    ok = bool(((info >= 1) & (info <= 4)).all())
    idx, val = res.index.cpu().numpy(), res.value.cpu().numpy()
    for i in (0, n // 2, n - 1):
        want = o.SingleProfileRestated(profs[i], edge_detection_method="Inflection Hill").inflection_data()
        got = [idx[i, 0], idx[i, 1], val[i, 0], val[i, 1]]
        ref = [want["left index (exact)"], want["right index (exact)"], want["left value (@exact)"], want["right value (@exact)"]]
        ok &= bool(np.allclose(got, ref, rtol=1e-5, atol=1e-5))
    entry("f4h", "SingleProfile(INFLECTION_HILL) constructor + inflection_data() for 4096 profiles x 200 detectors (x10 linear "
                 "resampling, BEAM_CENTER normalisation: 16 384 four-parameter Hill fits)", n, "profiles/s", dt,
          "two Hill penumbrae + dome + N(0, 0.05), numpy default_rng(4100)", {"units": 3, "ok": ok})
    out["f4h"]["nfev_mean"] = round(float(res.nfev.float().mean()), 1)
    # a Levenberg-Marquardt fit is bound by ~39 model evaluations of a 1 500-instruction pow() per sample, not by bytes:
    # no roofline fraction for this row, the rate in fits per second instead
    for k in ("algorithmic_GBs", "frac", "frac_of_measured_copy"):
        out["f4h"][k] = None
    out["f4h"]["fits_per_s"] = round(4 * n / dt, 1)
    out["f4h"]["bound"] = "compute (float64 pow per sample and function evaluation)"


def _median_rate(fn, warmup=3, repeats=5):
    """3 warm-up + 5 timed repeats, median -> (units per second, units per repeat, median seconds)"""
    for _ in range(warmup):
        fn()
    rs = sorted((fn() for _ in range(repeats)), key=lambda r: r[2])
    return rs[len(rs) // 2]


def cpu_baselines(frames_host, args, with_configs):
    """single-thread (3 warm-up + 5 repeats, median) + pool-of-cores oracle timings on bounded samples
    -> (headline cpu_baseline, per-config baselines)"""
    from oracle import cpu_baseline as cb
    from pylinac_amd.synthetic import pf_frames

    cores = os.cpu_count() or 1
    model = cb.cpu_model()
    n_sample = min(args.cpu_frames, frames_host.shape[0])
    rate, units, dt = _median_rate(lambda: cb.single_thread("epid", frames_host[:n_sample]))
    per_cfg = {}
    tasks = {"epid": [("epid", frames_host[i % frames_host.shape[0]][None]) for i in range(cores)]}
    singles = {}
    if with_configs:
        pf = pf_frames(8).numpy()
        wl_in, ct_in = cb._make_inputs("wl", 3000, 8), cb._make_inputs("ct", 4000, 8)
        singles["#3"] = _median_rate(lambda: cb.single_thread("pf", pf))
        singles["#4"] = _median_rate(lambda: cb.single_thread("wl", wl_in[0]))
        singles["#5"] = _median_rate(lambda: cb.single_thread("ct", ct_in[0], ct_in[1]))
        tasks["pf"] = [("pf", pf[i % 8][None]) for i in range(cores)]
        tasks["wl"] = [("wl", ("gen", 3000 + i, 2)) for i in range(cores)]
        tasks["ct"] = [("ct", ("gen", 4000 + i, 2)) for i in range(cores)]
    try:
        pool = cb.pool_throughput(tasks, cores)
    except Exception as exc:   # a baseline must not take the bench line down
        pool = {}
        print(f"[bench] pool baseline failed: {exc!r}", file=sys.stderr)
    ref = {}
    rpath = os.path.join(ROOT, "profiles", "cpu_reference.json")
    if os.path.exists(rpath):
        try:
            ref = json.load(open(rpath))
        except Exception:
            ref = {}
    head = {
        "value": round(rate, 3), "unit": "images/s", "cores": 1, "kind": "port",
        "sample": f"{units} of the same synthetic 1024x1024 uint16 frames through oracle/pylinac_oracle.py "
                  f"(scipy.ndimage gaussian+median, Otsu, threshold, np.mean, scipy.signal.find_peaks), single thread, "
                  f"3 warm-up + 5 repeats, median repeat {dt:.2f} s",
        "cpu_model": model, "host_cores": cores,
    }
    if "epid" in pool:
        head["pool"] = {"value": round(pool["epid"][0], 2), "cores": cores, "units": pool["epid"][1],
                        "wall_s": round(pool["epid"][2], 2),
                        "how": "multiprocessing spawn pool, one worker per host core (os.cpu_count()), one frame each, "
                               "wall clock from the first worker's start to the last worker's end"}
    if "#2" in ref:
        head["reference_kind"] = ref["#2"]          # the reference's own modules, timed in the build container (committed)
    names = {"#3": ("pf", "frames/s"), "#4": ("wl", "frames/s"), "#5": ("ct", "slices/s")}
    for key, (kind, unit) in names.items():
        if key not in singles:
            continue
        r, u, d = singles[key]
        per_cfg[key] = {"value": round(r, 3), "unit": unit, "cores": 1, "kind": "port",
                        "sample": f"{u} units through the oracle's restatement of the reference's per-image sequence, "
                                  f"3 warm-up + 5 repeats, median repeat {d:.2f} s"}
        if kind in pool:
            per_cfg[key]["pool"] = {"value": round(pool[kind][0], 2), "cores": cores, "units": pool[kind][1],
                                    "wall_s": round(pool[kind][2], 2)}
        if key in ref:
            per_cfg[key]["reference_kind"] = ref[key]
    if "#1" in ref:
        head["config_1_reference"] = ref["#1"]
    if "#4n" in ref and with_configs:
        per_cfg["#4n"] = dict(ref["#4n"], sample="the reference's own WLBaseImage sequence on 4 noisy frames, timed in the build "
                                                 "container (profiles/r03_cpu_reference.json); no live port timing for this variant")
    return head, per_cfg


def ensure_ranks(args):
    """`--gpus N` decides the job.  Under a launcher (WORLD_SIZE set) the two must agree -- a mismatch exits non-zero rather
    than timing a different job than the one asked for.  Without a launcher and N > 1 (plain `python bench.py --gpus 8`) this
    process starts the N ranks itself, exactly the way the driver's own command line does, and returns their exit code."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"[bench] --gpus {args.gpus} contradicts WORLD_SIZE={env_world}: refusing to time a different "
                             f"job than the one asked for")
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0]), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's only working mode on this host driver
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    ensure_ranks(args)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"[bench] rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # PL_BENCH_FORCE_DIST=1 exercises the RCCL init / all-gather / barrier path with a single rank too
    if world > 1 or os.environ.get("PL_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist  # noqa: F811

        dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"[bench] process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    from pylinac_amd import dist as pdist
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    h, w = args.height, args.width
    if args.scaling == "strong":
        lo, hi = pdist.shard_range(args.frames, rank, world)     # contiguous block split of the fixed frame set
        n, first, n_total = hi - lo, lo, args.frames
    else:
        n, first, n_total = args.frames, rank * args.frames, args.frames * world
    # global frame index -> seed 1000 + index
    frames = epid_open_field_frames(n, h, w, seed0=1000 + first, device=dev)
    pipe = EpidPipeline(n, h, w, dev)

    pending = []          # work handles of the record gathers in flight (N > 1): the gather of step k overlaps step k + 1

    def step(events=None):
        res = pipe.run(frames, events)
        rec = res.record()
        if dist is None:
            return rec
        while len(pending) >= 2:          # at most two gathers in flight: their output buffers stay bounded
            pending.pop(0).wait()
        return pdist.all_gather_records(rec, n_total, pending)

    def drain():
        while pending:
            pending.pop(0).wait()         # every gathered record table is complete before the clock stops

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    events = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = torch.empty(dist.get_world_size(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, t)              # each rank's own clock: a straggler GPU shows up by rank
        per_rank_ms = [round(float(x) / args.steps * 1e3, 4) for x in every.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-stage kernel time from the HIP events recorded INSIDE the timed region
    stage_ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in events.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    dom_s = stage_ms[dominant] / 1e3
    achieved = n * ALG_BYTES_PER_FRAME / dom_s / 1e9
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic, traffic_src = tj.get(dominant), tj.get("_measured_at")
        except Exception:
            traffic = None

    if rank == 0:
        total_frames = n_total * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        value = total_frames / elapsed
        line = {
            "metric": "EPID images/s, filter->threshold->profile->peak on 1024^2 batch",
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "rccl_ranks": (dist.get_world_size() if dist is not None else 0),   # ranks of the initialised process group
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 256 synthetic 1024x1024 uint16 EPID frames per GPU: "
                            "Gaussian(5)+median(3)+Otsu threshold on MI355X, then column-mean profile + FWXM peak",
                "frames_per_gpu": n, "frames_total": n_total, "height": h, "width": w,
                "parallelism": f"{world} x independent frame shards + 1 all-gather of [N,9] f64 records",
            },
            "roofline": {
                # "bound" names the roofline `achieved` / `peak` are priced on (BASELINE's metric: % of the HBM roofline, on
                # the algorithmic 4 MiB per frame); "limiter" names what actually holds the dominant kernel
                "bound": "hbm",
                "limiter": ("mfma+valu issue (HBM traffic is 1.06x algorithmic: profiles/pmc_traffic.json; DESIGN.md 9.1)"
                            if dominant == "gauss2d" else "hbm"),
                "kernel": dominant,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy": round(achieved / HBM_COPY_GBS, 4),
                "traffic": traffic,
                "traffic_measured_at": traffic_src,   # the build the PMC passes ran on (profiles/pmc_traffic.json)
                "note": "dominant kernel reads+writes one u16 frame (4 MiB/frame algorithmic); gauss2d = both Gaussian axes "
                        "in one launch, exact integer arithmetic on the matrix cores (v_mfma_i32_16x16x64_i8 over byte digit "
                        "planes), axis-0 plane kept in LDS; bound by MFMA + integer recombination issue, not by HBM "
                        "(DESIGN.md section 5)",
                # the same launch priced as matrix-core work (gauss2d only): 35 tiles of 16 x 16 outputs per 16 x 256
                # block, nine v_mfma_i32_16x16x64_i8 (32768 int8 ops) each.  Two peaks, named: the ~5 POP/s dense int8
                # figure of the data sheet, and the >= 3944 TOPS MI355X_MICROARCH.md MEASURED for this very instruction
                "mfma_view": (mfma_view(n * (h // 16) * (w // 256) * 35, dom_s)
                              if dominant == "gauss2d" and h % 16 == 0 and w % 256 == 0 else None),
                "pipeline_frac": round(value / world * ALG_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 4),
                "otsu_full_range_frames": int(pipe.flag.sum()),   # frames the one-pass window could not hold (0 on this workload)
                "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            },
        }
        if world == 1 and args.sustained_steps > 0:
            # The contract's W + K steps last ~25 ms: inside the power governor's ramp (an MI355X reaches its 1400 W cap under
            # gauss2d_mm; after an idle period the first launches run at boost clocks, the next ~10 are throttled hard, and the
            # clock settles over the following ~30 ms: profiles/r03_dvfs_ramp.txt).  `value` above is what the contract measures;
            # this object reports the same step after the governor has settled -- the rate of a production loop.
            for _ in range(max(args.sustained_steps // 3, 1)):
                step()
            drain()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(args.sustained_steps):
                step()
            drain()
            torch.cuda.synchronize()
            sdt = (time.perf_counter() - ts) / args.sustained_steps
            line["sustained"] = {"steps": args.sustained_steps, "settle_steps": max(args.sustained_steps // 3, 1),
                                 "ms_per_step": round(sdt * 1e3, 4), "value": round(n / sdt, 1), "unit": "images/s",
                                 "note": "same step, timed after the contract's region once clocks / power have settled; "
                                         "not the contract value"}
        if world == 1:
            # a sample of what the timed steps produced, against the CPU oracle (outside the timed region)
            if not args.no_parity:
                line["parity_sample"] = parity_epid(pipe.run(frames), frames)
            with_configs = not args.no_configs
            if with_configs:
                del pipe
                torch.cuda.empty_cache()
                cfg, keep = bench_configs(dev)
                line["configs"] = cfg
                del keep
            if not args.no_cpu_baseline:
                head, per_cfg = cpu_baselines(frames[: max(args.cpu_frames, 1)].cpu().numpy(), args, with_configs)
                line["cpu_baseline"] = head
                for key, val in per_cfg.items():
                    if key in line.get("configs", {}):
                        line["configs"][key]["cpu_baseline"] = val
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
