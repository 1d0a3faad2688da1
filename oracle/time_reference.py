"""TEST / BENCH INFRASTRUCTURE ONLY -- never imported by the product (pylinac_amd/).

`cpu_baseline.kind = "reference"` numbers: the REFERENCE'S OWN modules, imported read-only from
/root/reference through oracle/ref_loader.py, timed in the BUILD CONTAINER on the same seeded synthetic inputs bench.py
uses (configs #1 - #5), next to the oracle's restatement ("port") timed the same way in the same process, so the two kinds
can be compared.  /root/reference does not exist on the GPU box, so these numbers are committed
(profiles/r03_cpu_reference.json; profiles/cpu_reference.json = the compact form bench.py copies into its line) while
bench.py keeps timing the port live on the GPU box's host.

Timing: time.perf_counter, 3 warm-up + 5 repeats, median; single process / single thread, and a multiprocessing pool of
os.cpu_count() workers sharded by unit.  scikit-image is absent from this interpreter: the skimage-dependent sequences (#4,
#5) run under /opt/conda/bin/python3.9 (scikit-image 0.18.3, scipy 1.7.1; helper time_reference_py39.py); config #2's Otsu
step (the reference calls skimage.filters.threshold_otsu, pylinac/ct.py:3323) uses the oracle's bincount restatement.

    python oracle/time_reference.py            # ~3 min on 8 cores
"""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
PY39 = "/opt/conda/bin/python3.9"


def median_time(fn, warmup=3, repeats=5):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


# ------------------------------------------------------------------------------------------------ per-unit work (reference)
def _ref_modules():
    from oracle import ref_loader

    return ref_loader.ref("core.image"), ref_loader.ref("core.profile"), ref_loader.ref("picketfence")


def ref_epid(frames):
    """config #2 per frame, the reference's own calls: ArrayImage.filter(5, "gaussian") -> .filter(3, "median")
    (pylinac/core/image.py:695-712) -> Otsu -> .threshold(t) (:785-800) -> np.mean(array, 0)
    (pylinac/picketfence.py:747) -> FWXMProfile(...).field_edge_idx / center_idx / field_width_px
    (pylinac/core/profile.py:578-611)"""
    from oracle import pylinac_oracle as o

    image, profile, _ = _ref_modules()
    for f in frames:
        im = image.ArrayImage(f.copy())
        im.filter(size=5, kind="gaussian")
        im.filter(size=3, kind="median")
        t = o.threshold_otsu(im.array)
        im.threshold(t, kind="high")
        p = profile.FWXMProfile(np.mean(im.array, axis=0), fwxm_height=50)
        _ = (p.field_edge_idx("left"), p.field_edge_idx("right"), p.center_idx, p.field_width_px)
    return len(frames)


def ref_pf(frames, pixel_mm, sid, num_pickets):
    """configs #1 / #3: the constructor's ground() / normalize() (pylinac/picketfence.py:322-323) + the real
    PicketFence.analyze() (:636-845) on an array image"""
    image, _, pfm = _ref_modules()

    class PFImg(image.ArrayImage):
        _central_axis = None

        def adjust_for_sag(self, sag, orientation):
            pass

    for f in frames:
        im = PFImg(f.copy(), dpi=25.4 / pixel_mm, sid=sid)
        im.ground()
        im.normalize()
        pf = pfm.PicketFence(None)
        pf.image = im
        pf.analyze(orientation="Up-Down", num_pickets=num_pickets)
    return len(frames)


def port_epid(frames):
    from oracle import cpu_baseline as cb

    return cb._epid(frames)


def port_pf(frames, pixel_mm, sid, num_pickets):
    from oracle import pylinac_oracle as o

    dpmm = 1 / pixel_mm * sid / 1000
    for f in frames:
        o.pf_measure(o.normalize(o.ground(f)), dpmm, num_pickets=num_pickets)
    return len(frames)


_WORK = {"ref_epid": ref_epid, "ref_pf": ref_pf, "port_epid": port_epid, "port_pf": port_pf}


def _task(t):
    name, args = t
    t0 = time.time()
    _WORK[name](*args)
    return t0, time.time()


def pool_rate(name, unit_args, cores):
    """one unit per task, one worker per core, two rounds (the first warms every worker) -> units per second"""
    ctx = mp.get_context("spawn")
    tasks = [(name, unit_args[i % len(unit_args)]) for i in range(cores)]
    with ctx.Pool(cores) as pool:
        pool.map(_task, tasks, chunksize=1)
        rs = pool.map(_task, tasks * 2, chunksize=1)
    wall = max(r[1] for r in rs) - min(r[0] for r in rs)
    return len(rs) / wall


def main():
    from oracle import cpu_baseline as cb
    from pylinac_amd import synthetic

    cores = os.cpu_count() or 1
    out = {"_host": {"cpu_model": cb.cpu_model(), "cores": cores, "where": "build container (the GPU box has no /root/reference)",
                     "timing": "time.perf_counter, 3 warm-up + 5 repeats, median; pool = one worker per core, one unit per task"}}

    def both(key, what, units, ref_name, port_name, args_list, unit):
        r = {}
        for kind, name in (("reference", ref_name), ("port", port_name)):
            dt = median_time(lambda: _WORK[name](*args_list[0]))
            n = len(args_list[0][0])
            r[kind] = {"value": round(n / dt, 3), "unit": unit, "cores": 1, "sample": f"{n} units, median repeat {dt:.3f} s"}
            per_unit = [(a[0][i:i + 1],) + tuple(a[1:]) for a in args_list for i in range(len(a[0]))]
            r[kind]["pool"] = {"value": round(pool_rate(name, per_unit, cores), 2), "cores": cores}
        r["what"] = what
        out[key] = r
        print(key, json.dumps(r), flush=True)

    f1 = synthetic.pf_frames(2, 1280, 1280, seed0=1500, device="cpu", pixel_mm=0.224, pickets=9).numpy()
    both("#1", "PicketFence.analyze() on a synthetic 1280 x 1280 aS1200 frame at SID 1500 (0.336 mm pixels, 9 pickets): the "
         "CPU-runnable stand-in for PicketFence.from_demo_image().analyze() (pydicom and the demo file are absent)", 2,
         "ref_pf", "port_pf", [(f1, 0.336, 1500, 9)], "frames/s")
    f2 = synthetic.epid_open_field_frames(4, 1024, 1024, seed0=1000, device="cpu").numpy()
    both("#2", "Gaussian(5) + median(3) + Otsu + threshold + column profile + FWXM on 1024 x 1024 uint16 frames", 4,
         "ref_epid", "port_epid", [(f2,)], "images/s")
    f3 = synthetic.pf_frames(4, 768, 1024, seed0=2000, device="cpu").numpy()
    both("#3", "PicketFence.analyze() on 768 x 1024 aS1000 frames, 10 pickets, 60 leaf pairs", 4, "ref_pf", "port_pf",
         [(f3, 0.390625, 1000, 10)], "frames/s")

    # configs #4 / #5: scikit-image 0.18.3 lives under python3.9 -> helper process, same timing protocol inside it
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "i.npz"), os.path.join(td, "o.json")
        np.savez(inp, wl=synthetic.wl_frames(4, seed0=3000), wln=synthetic.wl_frames(4, seed0=3000, noise_sigma=0.001),
                 ct=synthetic.catphan_volume(4000, n_slices=16))
        subprocess.run([PY39, os.path.join(HERE, "time_reference_py39.py"), inp, outp, ROOT], check=True)
        out.update(json.load(open(outp)))
    wl_in = synthetic.wl_frames(4, seed0=3000)
    wln_in = synthetic.wl_frames(4, seed0=3000, noise_sigma=0.001)
    vol = synthetic.catphan_volume(4000, n_slices=16)
    for key, fn in (("#4", lambda: cb._wl(wl_in)), ("#4n", lambda: cb._wl(wln_in)), ("#5", lambda: cb._ct(vol, list(range(3, 13))))):
        dt = median_time(fn)
        n = 4 if key != "#5" else 10
        out[key]["port"] = {"value": round(n / dt, 3), "cores": 1, "sample": f"{n} units, median repeat {dt:.3f} s",
                            "interpreter": "python 3.10 / scipy 1.15.3 (the oracle's restatement of the scikit-image steps)"}
        print(key, json.dumps(out[key]), flush=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_cpu_reference.json"), "w"), indent=1)
    compact = {k: {"value": v["reference"]["value"], "unit": v["reference"].get("unit", ""), "cores": 1, "kind": "reference",
                   "pool": v["reference"].get("pool"), "port_same_host": v.get("port", {}).get("value"),
                   "host": out["_host"]["cpu_model"] + f", {cores} cores, build container"}
               for k, v in out.items() if not k.startswith("_")}
    json.dump(compact, open(os.path.join(ROOT, "profiles", "cpu_reference.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
