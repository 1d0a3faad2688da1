#!/usr/bin/env python
"""Generates tests/golden/*.npz by running THE REFERENCE'S OWN CODE (read-only from
/root/reference through oracle/ref_loader.py, real scipy 1.15.3 underneath) and, for the
scikit-image bits, scikit-image 0.18.3 under /opt/conda/bin/python3.9 (helper: skimage_py39.py).

Run in the build container only:   python tests/golden/make_golden.py
The .npz files are committed; the GPU box (which has no /root/reference) only reads them.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402

PY39 = "/opt/conda/bin/python3.9"


def synth_frames(n, h, w, seed, dtype=np.uint16):
    """Small EPID-like frames: blurred square field + noise + dead/hot pixels (numpy, seeded)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        cy, cx = h / 2 + rng.uniform(-3, 3), w / 2 + rng.uniform(-3, 3)
        hy, hx = h * 0.28, w * 0.3
        s = 2.5
        from scipy.special import erf

        fy = 0.5 * (erf((y - (cy - hy)) / s) - erf((y - (cy + hy)) / s))
        fx = 0.5 * (erf((x - (cx - hx)) / s) - erf((x - (cx + hx)) / s))
        img = 2000 + 38000 * fy * fx + rng.normal(0, 400, (h, w))
        pos = rng.integers(0, h * w, 6)
        img.ravel()[pos] = rng.choice([0, 65535], 6)
        out.append(np.clip(np.round(img), 0, 65535))
    a = np.stack(out)
    if dtype == np.int16:
        return (a - 32768).astype(np.int16)
    return a.astype(dtype)


def ndimage_gaussian(a, s):
    from scipy import ndimage as ndi

    return ndi.gaussian_filter(a, s)


def ring_mask(h, w):
    y, x = np.mgrid[0:h, 0:w]
    r = np.hypot(y - h / 2, x - w / 2)
    m = ((r > 10) & (r < 18)) | ((r > 25) & (r < 33)) | (r < 4)
    m[5:12, 5:12] = 1
    m[7:10, 7:10] = 0      # a hole away from the rings
    m[0:6, 60:70] = 1      # touches the border
    m[1:5, 62:68] = 0      # enclosed hole inside a border-touching blob
    m[0, 64] = 0           # ...opened to the border through one pixel
    return m.astype(np.uint8)


def wl_frames(n, h, w, seed):
    """WL-like frames: blurred 20 mm-ish field with a dark BB near the centre + noise."""
    from scipy import ndimage as ndi

    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        img = np.full((h, w), 1500.0)
        cy, cx = h / 2 + rng.uniform(-6, 6), w / 2 + rng.uniform(-6, 6)
        img[int(cy - 30):int(cy + 30), int(cx - 30):int(cx + 30)] = 42000.0
        y, x = np.mgrid[0:h, 0:w]
        bb = np.hypot(y - (cy + rng.uniform(-3, 3)), x - (cx + rng.uniform(-3, 3))) < 7.5
        img[bb] *= 0.2
        img = ndi.gaussian_filter(img, 2.0) + rng.normal(0, 150, (h, w))
        out.append(np.clip(np.round(img), 0, 65535))
    return np.stack(out).astype(np.uint16)


def catphan_slices(n, size, seed):
    """CatPhan-like int16 HU slices: air (-1000), a 200 mm acrylic-ish cylinder (~90 HU) slightly off
    centre, a few inserts and air holes, a couch bar that touches the image border, noise."""
    rng = np.random.default_rng(seed)
    mmpp = 250.0 / size
    out = []
    for i in range(n):
        y, x = np.mgrid[0:size, 0:size].astype(float)
        cy, cx = size / 2 + rng.uniform(-6, 6), size / 2 + rng.uniform(-6, 6)
        r = np.hypot(y - cy, x - cx) * mmpp
        img = np.full((size, size), -1000.0)
        img[r < 100] = 90.0
        for k in range(6):
            a = k * np.pi / 3 + 0.2 * i
            iy, ix = cy + 58 / mmpp * np.sin(a), cx + 58 / mmpp * np.cos(a)
            img[np.hypot(y - iy, x - ix) * mmpp < 6] = [-1000, 340, -200, 950, -100, 120][k]
        img[int(size * 0.975):int(size * 0.995), :] = 200.0        # couch: touches the border band
        img += rng.normal(0, 12, img.shape)
        out.append(np.round(img))
    return np.stack(out).astype(np.int16), mmpp


def pf_frame(h, w, pixel, seed, n_pickets=10, spacing_mm=15.0, gap_mm=2.0):
    """Synthetic UP_DOWN picket fence (config #3 restated): Gaussian pickets with N(0, 0.2 mm) offsets,
    one closed leaf pair region (jaw-blocked rows) and noise."""
    from scipy import ndimage as ndi

    rng = np.random.default_rng(seed)
    x = (np.arange(w) - (w - 1) / 2) * pixel
    centers = (np.arange(n_pickets) - (n_pickets - 1) / 2) * spacing_mm + rng.normal(0, 0.2, n_pickets)
    prof = np.zeros(w)
    for c in centers:
        prof += np.exp(-0.5 * ((x - c) / (gap_mm / 2.355 * 1.6)) ** 2)
    img = 1000.0 + 30000.0 * np.repeat(prof[None, :], h, axis=0)
    img[: int(h * 0.06), :] = 1000.0                       # rows hidden by a jaw: windows there must be rejected
    img = ndi.gaussian_filter(img, 1.0) + rng.normal(0, 60, (h, w))
    return np.clip(np.round(img), 0, 65535).astype(np.uint16)


def bb_windows(n, seed):
    """The ~134 x 134 px search window of WL's SizedDiskLocator ((40 + 5) mm at 2.98 dpmm,
    pylinac/winston_lutz.py:795-805): normalised float64 field plateau with a dark 5 mm BB, mild blur,
    noise, and (window 2) a thin rod attached to the BB."""
    from scipy import ndimage as ndi

    rng = np.random.default_rng(seed)
    dpmm = 1 / 0.336
    size = int(np.ceil(45 * dpmm))
    out = []
    for i in range(n):
        y, x = np.mgrid[0:size, 0:size].astype(float)
        cy, cx = size / 2 + rng.uniform(-8, 8), size / 2 + rng.uniform(-8, 8)
        img = np.full((size, size), 0.92)
        img[np.hypot(y - cy, x - cx) < 2.5 * dpmm] = 0.35
        if i == 2:
            img[int(cy):, int(cx) - 1:int(cx) + 2] = 0.5       # BB rod: spiculated region
        if i in (4, 5):   # a second BB 8 mm (i == 4) / 20 mm (i == 5) to the right: same-level duplicates
            sep = (8 if i == 4 else 20) * dpmm
            cx = size / 2 - sep / 2
            cy = size / 2
            img = np.full((size, size), 0.92)
            img[np.hypot(y - cy, x - cx) < 2.5 * dpmm] = 0.35
            img[np.hypot(y - cy - 1.3, x - cx - sep) < 2.5 * dpmm] = 0.35
        img = ndi.gaussian_filter(img, 1.2) + rng.normal(0, 0.004, img.shape)
        out.append(img)
    return out, dpmm


def field_frames(seed):
    """Multi-target WL style frames (float64, normalised): several small square fields on a dark background,
    one of them touching the border band, one of the wrong size, one with an interior BB shadow (a hole at high
    thresholds), mild blur + noise.  Returns frames, dpmm and per-frame locator arguments."""
    from scipy import ndimage as ndi

    rng = np.random.default_rng(seed)
    dpmm = 1 / 0.336
    frames, args = [], []
    for i in range(3):
        h, w = (420, 560) if i != 2 else (300, 300)
        img = np.full((h, w), 0.04)
        side = 15 * dpmm
        centres = [(110, 130), (300, 150), (200, 400)] if i != 2 else [(150, 150)]
        if i == 1:
            centres += [(int(side / 2) + 2, 300)]                   # inside the clear_border band -> ignored
            img[330:330 + int(8 * dpmm), 450:450 + int(8 * dpmm)] = 0.9   # 8 mm field: wrong size
        for (cy, cx) in centres:
            cy, cx = cy + rng.uniform(-0.5, 0.5), cx + rng.uniform(-0.5, 0.5)
            y, x = np.mgrid[0:h, 0:w].astype(float)
            sq = (np.abs(y - cy) < side / 2) & (np.abs(x - cx) < side / 2)
            img[sq] = 0.8 + 0.15 * rng.uniform()
            img[np.hypot(y - cy - 2, x - cx + 3) < 2.5 * dpmm] *= 0.55     # BB shadow inside the field
        img = ndi.gaussian_filter(img, 1.5) + rng.normal(0, 0.006, img.shape)
        frames.append(img)
        args.append(dict(fw=15.0, fh=15.0, tol=3.0 if i != 2 else 1.5, maxn=3 if i != 2 else 1))
    return frames, dpmm, args


def skimage_otsu(arrays: dict) -> dict:
    """threshold_otsu via scikit-image 0.18.3 in the py3.9 interpreter."""
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "in.npz")
        outp = os.path.join(td, "out.json")
        np.savez(inp, **arrays)
        subprocess.run([PY39, os.path.join(HERE, "skimage_py39.py"), inp, outp], check=True,
                       stderr=subprocess.DEVNULL)
        return json.load(open(outp))


def main():
    au = ref_loader.ref("core.array_utils")
    prof = ref_loader.ref("core.profile")
    image = ref_loader.ref("core.image")
    meta = {
        "generator": "tests/golden/make_golden.py",
        "reference": "pylinac 3.46.0 (/root/reference, read-only)",
        "numpy": np.__version__,
        "scipy": __import__("scipy").__version__,
        "skimage": "0.18.3 (py3.9 helper)",
    }

    # ---------------------------------------------------------------- 1. frame filters / mutators
    g = {}
    frames = synth_frames(3, 96, 128, seed=11)
    rnd = np.random.default_rng(5).integers(0, 65536, (2, 64, 72), dtype=np.uint16)
    small = np.random.default_rng(6).integers(0, 65536, (1, 7, 9), dtype=np.uint16)
    sets = {"field": frames, "random": rnd, "tiny": small}
    for name, fs in sets.items():
        g[f"{name}.in"] = fs
        for tag, size, kind in [("g5", 5, "gaussian"), ("g1", 1, "gaussian"), ("g2", 2, "gaussian"), ("gf03", 0.03, "gaussian"),
                                ("m3", 3, "median"), ("m5", 5, "median"), ("m2", 2, "median"),
                                ("mf05", 0.05, "median")]:
            outs = []
            for f in fs:
                im = image.ArrayImage(f.copy())
                im.filter(size=size, kind=kind)  # pylinac/core/image.py:695-712
                outs.append(im.array)
            g[f"{name}.filter.{tag}"] = np.stack(outs)
        outs_hi, outs_lo, outs_bin, outs_gr, outs_no, outs_inv = [], [], [], [], [], []
        for f in fs:
            im = image.ArrayImage(f.copy()); im.threshold(30000); outs_hi.append(im.array)
            im = image.ArrayImage(f.copy()); im.threshold(30000, kind="low"); outs_lo.append(im.array)
            outs_bin.append(image.ArrayImage(f.copy()).as_binary(30000).array)
            im = image.ArrayImage(f.copy()); im.ground(); outs_gr.append(im.array)
            im = image.ArrayImage(f.copy()); im.normalize(); outs_no.append(im.array)
            im = image.ArrayImage(f.copy()); im.invert(); outs_inv.append(im.array)
        g[f"{name}.threshold.high"] = np.stack(outs_hi)
        g[f"{name}.threshold.low"] = np.stack(outs_lo)
        g[f"{name}.as_binary"] = np.stack(outs_bin)
        g[f"{name}.ground"] = np.stack(outs_gr)
        g[f"{name}.normalize"] = np.stack(outs_no)
        g[f"{name}.invert"] = np.stack(outs_inv)
        g[f"{name}.stretch"] = np.stack([au.stretch(f.astype(float), 0, 1) for f in fs])
        g[f"{name}.percentiles"] = np.stack([np.percentile(f, [0.5, 5, 50, 95, 99.5, 99.9]) for f in fs])
    # float and int16 frames
    ff = (frames[:2].astype(np.float64) / 65535.0)
    g["float64.in"] = ff
    g["float64.filter.g2"] = np.stack([au.filter(f, 2, "gaussian") for f in ff])
    g["float64.filter.m3"] = np.stack([au.filter(f, 3, "median") for f in ff])
    f32 = ff.astype(np.float32)
    g["float32.in"] = f32
    g["float32.filter.g2"] = np.stack([au.filter(f, 2, "gaussian") for f in f32])
    i16 = synth_frames(2, 64, 80, seed=3, dtype=np.int16)
    g["int16.in"] = i16
    g["int16.filter.g2"] = np.stack([au.filter(f, 2, "gaussian") for f in i16])
    g["int16.filter.m3"] = np.stack([au.filter(f, 3, "median") for f in i16])
    # reference KATs (tests_basic/core/test_array_utils.py:65-149, test_image.py:450-461, 522-535)
    kat = np.array([0, 0, 0, 3, 0, 0, 0])
    g["kat.filter.in"] = kat
    g["kat.filter.median1"] = au.filter(kat, size=1, kind="median")
    g["kat.filter.median_f01"] = au.filter(kat, size=0.1, kind="median")
    g["kat.filter.median3"] = au.filter(kat, size=3, kind="median")
    g["kat.filter.median3b"] = au.filter(np.array([0, 0, 3, 3, 0, 0, 0]), size=3, kind="median")
    g["kat.filter.gauss1"] = au.filter(kat, size=1, kind="gaussian")
    im = image.ArrayImage(np.arange(42).reshape(6, 7)); im.filter(3); g["kat.image.filter3"] = im.array
    im = image.ArrayImage(np.arange(42).reshape(6, 7)); im.threshold(10); g["kat.image.threshold10"] = im.array
    im = image.ArrayImage(np.arange(42).reshape(6, 7)); im.threshold(20, kind="low"); g["kat.image.threshold20low"] = im.array
    g["kat.normalize"] = au.normalize(np.array((1, 2, 3, 4)))
    g["kat.normalize2"] = au.normalize(np.array((1, 2, 3, 4), dtype=float), 2)
    g["kat.invert"] = au.invert(np.array([0, 10]))
    g["kat.invert_neg"] = au.invert(np.array([-5, -1]))
    g["kat.ground"] = au.ground(np.array([3, 4, 5]))
    g["kat.ground_neg"] = au.ground(np.array([-3, -4, -5]))
    g["kat.ground10"] = au.ground(np.array([3, 4, 5]), value=10)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **g)

    # -------------------------------------------------------------------------- 2. Otsu (skimage)
    o = {}
    o["u16_field"] = np.stack([au.filter(f, 3, "gaussian") for f in frames])
    o["u16_random"] = rnd
    o["i16"] = i16
    o["const"] = np.full((1, 16, 16), 1234, dtype=np.uint16)
    o["two_level"] = np.where(np.arange(400).reshape(1, 20, 20) % 3 == 0, 100, 900).astype(np.uint16)
    flat = {}
    for k, v in o.items():
        for i, f in enumerate(v):
            flat[f"{k}.{i}"] = f
    thr = skimage_otsu(flat)
    og = {f"{k}.in": v for k, v in o.items()}
    for k, v in o.items():
        og[f"{k}.otsu"] = np.array([thr[f"{k}.{i}"] for i in range(len(v))], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "otsu.npz"), **og)

    # ---------------------------------------------------------------------------------- 3. peaks
    pk = {}
    rng = np.random.default_rng(21)
    import scipy.signal as sps

    profiles = {
        "simple9": np.array([0, 1, 2, 3, 4, 3, 2, 1, 0], dtype=float),
        "simple8": np.array([0, 1, 2, 3, 3, 2, 1, 0], dtype=float),
        "long23": np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0], dtype=float),
        "long22": np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0], dtype=float),
        "skewed19": np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 8, 6, 4, 2, 0], dtype=float),
        "sigmoid21": np.array([0, 1, 2, 4, 6, 8, 9, 10, 10, 10, 10, 10, 10, 10, 9, 8, 6, 4, 2, 1, 0], dtype=float),
        "sawtooth": sps.sawtooth(np.linspace(0, 8 * np.pi, num=200), width=0.5),
        "walk600": np.abs(rng.normal(size=600).cumsum()),
        "pickets": sum(np.exp(-0.5 * ((np.arange(1200) - c) / 6.0) ** 2) for c in np.arange(100, 1150, 110))
                   + rng.normal(0, 0.01, 1200) + 0.05,
        "noisy_field": np.convolve(np.r_[np.zeros(150), np.ones(500), np.zeros(150)], np.ones(25) / 25, "same")
                       + rng.normal(0, 0.004, 800),
    }
    variants = {
        "default": dict(),
        "fwxm_top1": dict(fwxm_height=0.5, max_number=1),
        "fwxm25_top1": dict(fwxm_height=0.25, max_number=1),
        "fwxm75_top1": dict(fwxm_height=0.75, max_number=1),
        "mp_default": dict(threshold=0.3, peak_separation=0.05),
        "pf": dict(threshold=0.5, peak_separation=0.02, peak_sort="peak_heights", required_prominence=0.2),
        "region": dict(threshold=0.3, peak_separation=0.05, search_region=(0.2, 0.8), max_number=2),
        "region_int": dict(threshold=0.1, peak_separation=5, search_region=(3, 190)),
    }
    names = []
    for pname, vals in profiles.items():
        pk[f"{pname}.values"] = vals
        for vname, kw in variants.items():
            try:
                idx, props = prof.find_peaks(vals.copy(), **kw)  # pylinac/core/profile.py:2545
            except Exception as e:  # noqa: BLE001
                pk[f"{pname}.{vname}.error"] = np.array(type(e).__name__)
                continue
            names.append(f"{pname}.{vname}")
            pk[f"{pname}.{vname}.idx"] = idx
            for k, v in props.items():
                pk[f"{pname}.{vname}.{k}"] = v
        mp = prof.MultiProfile(vals.copy())
        a, b = mp.find_peaks(); pk[f"{pname}.mp.peaks.idx"], pk[f"{pname}.mp.peaks.val"] = a, b
        a, b = mp.find_valleys(); pk[f"{pname}.mp.valleys.idx"], pk[f"{pname}.mp.valleys.val"] = a, b
        a, b = mp.find_fwxm_peaks(); pk[f"{pname}.mp.fwxm.idx"], pk[f"{pname}.mp.fwxm.val"] = a, b
        for hgt in (25, 50, 75):
            try:
                fp = prof.FWXMProfile(vals.copy(), fwxm_height=hgt)
                pk[f"{pname}.fwxm{hgt}"] = np.array([fp.field_edge_idx("left"), fp.field_edge_idx("right"),
                                                     fp.center_idx, fp.field_width_px])
            except Exception as e:  # noqa: BLE001
                pk[f"{pname}.fwxm{hgt}.error"] = np.array(type(e).__name__)
    pk["variants"] = np.array(json.dumps(variants))
    np.savez_compressed(os.path.join(HERE, "peaks.npz"), **pk)

    # -------------------------------------------------- 4. composed EPID pipeline (config #2 + peak)
    ep = {}
    pf = synth_frames(3, 128, 160, seed=77)
    ep["in"] = pf
    med = []
    for f in pf:
        im = image.ArrayImage(f.copy())
        im.filter(5, "gaussian")
        im.filter(3, "median")
        med.append(im.array)
    med = np.stack(med)
    thr = skimage_otsu({str(i): m for i, m in enumerate(med)})
    ts = np.array([thr[str(i)] for i in range(len(med))], dtype=np.int64)
    outs, profs, recs = [], [], []
    for m, t in zip(med, ts):
        im = image.ArrayImage(m.copy())
        im.threshold(int(t))
        outs.append(im.array)
        p = np.mean(im.array, axis=0)  # pylinac/picketfence.py:747-750
        profs.append(p)
        fp = prof.FWXMProfile(p, fwxm_height=50)
        recs.append([fp.field_edge_idx("left"), fp.field_edge_idx("right"), fp.center_idx, fp.field_width_px])
    ep["median"] = med
    ep["otsu"] = ts
    ep["out"] = np.stack(outs)
    ep["profile"] = np.stack(profs)
    ep["fwxm"] = np.array(recs)
    np.savez_compressed(os.path.join(HERE, "epid_pipeline.npz"), **ep)

    # ------------------------------------------- 5. labels (skimage), circle profiles, Sobel, WL field
    rng = np.random.default_rng(8)
    masks = {
        "random40": (rng.random((40, 56)) > 0.55).astype(np.uint8),
        "random_dense": (rng.random((33, 47)) > 0.3).astype(np.uint8),
        "blobs": (ndimage_gaussian(rng.random((96, 128)), 3) > 0.5).astype(np.uint8),
        "rings": ring_mask(90, 110),
        "empty": np.zeros((8, 9), np.uint8),
        "full": np.ones((7, 5), np.uint8),
    }
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "m.npz"), os.path.join(td, "l.npz")
        np.savez(inp, **masks)
        subprocess.run([PY39, os.path.join(HERE, "skimage_label_py39.py"), inp, outp], check=True,
                       stderr=subprocess.DEVNULL)
        lab = dict(np.load(outp))
    from scipy import ndimage as ndi

    misc = {f"mask.{k}": v for k, v in masks.items()}
    for k, v in lab.items():
        misc[f"label.{k}"] = v
    for k, m in masks.items():
        misc[f"fill.{k}"] = ndi.binary_fill_holes(m).astype(np.uint8)   # winston_lutz.py:777
    # circle profiles through the reference classes (pylinac/core/profile.py:2179-2483)
    geo = ref_loader.ref("core.geometry")
    img16 = synth_frames(1, 160, 200, seed=5)[0]
    imgf = img16.astype(float) / 65535.0
    misc["circle.img16"] = img16
    cases = [(100.3, 80.7, 50.0, 0.0, True, 1.0), (90.0, 70.0, 60.2, 0.7, False, 2.0), (120.0, 85.0, 70.0, 0.0, True, 1.5)]
    misc["circle.cases"] = np.array(cases)
    for i, (cx, cy, r, sa, ccw, sr) in enumerate(cases):
        cp = prof.CircleProfile(geo.Point(cx, cy), r, img16, start_angle=sa, ccw=bool(ccw), sampling_ratio=sr)
        misc[f"circle.{i}.u16"] = np.asarray(cp.values)
        cp = prof.CircleProfile(geo.Point(cx, cy), r, imgf, start_angle=sa, ccw=bool(ccw), sampling_ratio=sr)
        misc[f"circle.{i}.f64"] = np.asarray(cp.values)
        ccp = prof.CollapsedCircleProfile(geo.Point(cx, cy), r, imgf, start_angle=sa, ccw=bool(ccw), sampling_ratio=sr,
                                          width_ratio=0.1, num_profiles=20)
        misc[f"collapsed.{i}.f64"] = np.asarray(ccp.values)
        ccp = prof.CollapsedCircleProfile(geo.Point(cx, cy), r, img16, sampling_ratio=sr, width_ratio=0.05, num_profiles=5)
        misc[f"collapsed.{i}.u16"] = np.asarray(ccp.values)
    # Sobel as BaseImage.gamma calls it (pylinac/core/image.py:1006-1007): float32
    f32img = imgf.astype(np.float32)
    misc["sobel.in"] = f32img
    misc["sobel.axis1"] = ndi.sobel(f32img, 1)
    misc["sobel.axis0"] = ndi.sobel(f32img, 0)
    # WL field centroid on synthetic WL frames (field + BB): reference expressions of
    # winston_lutz.py:711-712, 775-779 via the reference's ArrayImage
    wl = wl_frames(3, 200, 240, seed=31)
    misc["wl.in"] = wl
    cen = []
    for f in wl:
        im = image.ArrayImage(f.copy())
        im.ground()
        im.normalize()
        mn, mx = np.percentile(im.array, [5, 99.9])
        thr_img = im.as_binary((mx - mn) / 2 + mn)
        filled = ndi.binary_fill_holes(thr_img.array)
        c = ndi.center_of_mass(filled)
        cen.append([c[-1], c[0], filled.sum()])
    misc["wl.centroid"] = np.array(cen)
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc)

    # ------------------------------------------ 6. CatPhan slice localisation (skimage 0.18.3, py3.9)
    slices, mmpp = catphan_slices(3, 256, seed=17)
    from scipy import ndimage as ndi2

    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "s.npz"), os.path.join(td, "o.npz")
        common = dict(slices=slices, mm_per_pixel=mmpp, catphan_size=np.pi * 101**2 / mmpp**2)
        np.savez(inp, stage="scharr", **common)
        subprocess.run([PY39, os.path.join(HERE, "skimage_ct_py39.py"), inp, outp], check=True,
                       stderr=subprocess.DEVNULL)
        sch = dict(np.load(outp))
        gauss = {f"gauss{i}": ndi2.gaussian_filter(sch[f"{i}.scharr"], 1, mode="nearest", truncate=4.0)
                 for i in range(len(slices))}          # skimage.filters.gaussian defaults, scipy 1.15.3
        np.savez(inp, stage="rest", **common, **gauss)
        subprocess.run([PY39, os.path.join(HERE, "skimage_ct_py39.py"), inp, outp], check=True,
                       stderr=subprocess.DEVNULL)
        ctg = dict(np.load(outp))
        ctg.update(sch)
    for i in (1, 2):   # the float stages of one slice are enough to pin the oracle; keep the file small
        ctg.pop(f"{i}.scharr"); ctg.pop(f"{i}.gauss")
    ctg["slices"] = slices
    ctg["mm_per_pixel"] = np.float64(mmpp)
    ctg["catphan_size"] = np.float64(np.pi * 101**2 / mmpp**2)   # pylinac/ct.py:2581-2584 (radius 101 mm)
    # float64 is kept for the two float stages; the masks/labels are small once compressed
    np.savez_compressed(os.path.join(HERE, "catphan.npz"), **ctg)

    # ----------------------------- 7. picket fence: the reference's REAL PicketFence.analyze() (config #3)
    pfm = ref_loader.ref("picketfence")

    class PFImg(image.ArrayImage):          # what PFDicomImage adds to the array image (picketfence.py:204-260)
        _central_axis = None

        def adjust_for_sag(self, sag, orientation):
            pass

    pfg = {}
    for k, (hh, ww, pixel, seed) in enumerate([(400, 520, 0.78125, 2000), (384, 512, 0.8, 2001)]):
        raw = pf_frame(hh, ww, pixel, seed)
        dpmm = 1 / pixel
        im = PFImg(raw.copy(), dpi=dpmm * 25.4, sid=1000)
        im.crop(pixels=int(round(3 * im.dpmm)))             # picketfence.py:214-215
        cropped = np.ascontiguousarray(im.array)
        im.ground()
        im.normalize()                                       # picketfence.py:322-323
        pf = pfm.PicketFence(None)                           # skips image loading (picketfence.py:315)
        pf.image = im
        pf.analyze(orientation="Up-Down")
        pfg[f"{k}.cropped"] = cropped
        pfg[f"{k}.dpmm"] = np.float64(im.dpmm)
        pfg[f"{k}.meas"] = np.array([[m.leaf_num, m.picket_num, m.position[0], m._approximate_idx] for m in pf.mlc_meas])
        pfg[f"{k}.spacing"] = np.float64(pf.mlc_meas[0]._spacing)
        pfg[f"{k}.max_error"] = np.float64(pf.max_error)
    np.savez_compressed(os.path.join(HERE, "picketfence.npz"), **pfg)

    # ----------------- 8. BB finder: the reference's own find_features under scikit-image 0.18.3 (a13)
    bbw, bb_dpmm = bb_windows(6, seed=51)
    bb_maxn, bb_minsep = [1, 1, 1, 1, 3, 2], [5, 5, 5, 5, 12, 5]   # windows 4/5: twin BBs
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "w.npz"), os.path.join(td, "f.npz")
        np.savez(inp, count=len(bbw), dpmm=bb_dpmm, radius_mm=2.5, tol_mm=0.5, maxn=bb_maxn, minsep=bb_minsep,
                 **{f"w{i}": w for i, w in enumerate(bbw)})
        subprocess.run([PY39, os.path.join(HERE, "skimage_features_py39.py"), inp, outp, ROOT], check=True)
        bbg = dict(np.load(outp))
    for i, w in enumerate(bbw):
        bbg[f"{i}.window"] = w
    bbg["dpmm"] = np.float64(bb_dpmm)
    bbg["maxn"], bbg["minsep"] = np.array(bb_maxn), np.array(bb_minsep, dtype=float)
    np.savez_compressed(os.path.join(HERE, "features.npz"), **bbg)

    # ------------------- 9. NPS / radial average / ESF-FFT MTF: the reference's own functions (a18)
    nps = ref_loader.ref("core.nps")
    rmtf = ref_loader.ref("core.mtf")
    from scipy.signal import windows

    def noisy(shape, scale, intensity, seed=123):      # tests_basic/core/test_nps.py:14-80, seeded the same way
        rng = np.random.default_rng(seed=seed)
        low = rng.normal(loc=0, scale=intensity, size=(shape[0] // scale, shape[1] // scale))
        m = np.kron(low, np.ones((scale, scale)))[: shape[0], : shape[1]]
        return np.clip(np.zeros(shape, dtype=np.uint16) + m, 0, 65535)

    sp = {}
    roi1, roi2, roi3 = noisy((300, 300), 30, 500), noisy((200, 200), 10, 100), noisy((65, 50), 5, 80, seed=9)[:61, :47]
    hu = np.random.default_rng(77).normal(40, 12, (4, 45, 45))          # CT-like uniformity ROIs (float)
    sp.update(roi1=roi1, roi2=roi2, roi3=roi3, hu=hu)
    sp["nps_single"] = nps.noise_power_spectrum_2d(pixel_size=1, rois=[roi1])
    sp["nps_two"] = nps.noise_power_spectrum_2d(pixel_size=0.5, rois=[roi1, roi2])
    sp["nps_ragged"] = nps.noise_power_spectrum_2d(pixel_size=0.39, rois=[roi2, roi3])
    sp["nps_odd"] = nps.noise_power_spectrum_2d(pixel_size=1, rois=[roi1[:-1, :-1]])
    sp["nps_hu"] = nps.noise_power_spectrum_2d(pixel_size=0.48, rois=list(hu))
    for k in ("single", "two", "ragged", "odd", "hu"):
        one = nps.noise_power_spectrum_1d(sp[f"nps_{k}"])
        sp[f"nps1d_{k}"] = one
        sp[f"scalars_{k}"] = np.array([nps.average_power(one), nps.max_frequency(one)])
    sp["radial_ones"] = nps.radial_average(np.ones((300, 300)))
    rect = np.random.default_rng(3).normal(5, 2, (37, 64))
    sp["rect"], sp["radial_rect"] = rect, nps.radial_average(rect)
    # ESF-FFT MTF: the reference's KAT inputs (tests_basic/core/test_mtf.py:59-132) + a blurred noisy edge
    step = lambda n: np.append(np.zeros(n // 2), np.ones(n // 2))  # noqa: E731
    shifted = np.zeros(256)
    shifted[228:] = 1
    from scipy.special import erf

    rng = np.random.default_rng(8)
    blur = [0.5 * (1 + erf((np.arange(n) - n / 2 + d) / 3.0)) * 900 + 50 + rng.normal(0, 2, n)
            for n, d in ((120, 0.3), (97, -1.2), (150, 2.0))]
    cases = {
        "single": dict(esf=[step(8)]),
        "multi": dict(esf=[step(8), step(6)]),
        "spacing": dict(esf=[step(8), step(6)], sample_spacing=10),
        "kaiser": dict(esf=[step(8), step(6)], windowing=windows.kaiser, beta=0.5),
        "shift_none": dict(esf=[shifted], windowing=None),
        "shift_hann": dict(esf=[shifted]),
        "shift_tukey": dict(esf=[shifted], windowing=windows.tukey, alpha=0.2),
        "pad_none": dict(esf=[step(256), step(256)], padding_mode="none"),
        "pad_fixed": dict(esf=[step(8), step(6)], padding_mode="fixed", num_samples=100),
        "blur": dict(esf=blur, sample_spacing=0.25),
    }
    for name, kw in cases.items():
        m = rmtf.EdgeSpreadFunctionMTF(**kw)
        sp[f"esf_{name}.freq"], sp[f"esf_{name}.mtf"] = m.freq, m.mtf
        sp[f"esf_{name}.each"] = np.array(m._mtf)
        sp[f"esf_{name}.res"] = np.array([m.relative_resolution(t) for t in (30, 50, 80)])
        for i, e in enumerate(kw["esf"]):
            sp[f"esf_{name}.in{i}"] = e
    np.savez_compressed(os.path.join(HERE, "spectral.npz"), **sp)

    # ---- 10. SingleProfile (a11): the reference's 20 frozen profile fixtures x 6 resampling modes + EPID cases
    import importlib.util

    fa = ref_loader.ref("field_analysis")
    spec = importlib.util.spec_from_file_location(
        "profile_regression_fixtures", "/root/reference/tests_basic/core/profile_regression_fixtures.py")
    fxm = importlib.util.module_from_spec(spec)
    sys.modules["profile_regression_fixtures"] = fxm
    spec.loader.exec_module(fxm)
    calcs = {"varian_flatness_difference": fa.flatness_dose_difference,
             "varian_symmetry_point_difference": fa.symmetry_point_difference,
             "elekta_flatness_ratio": fa.flatness_dose_ratio, "elekta_symmetry_pdq": fa.symmetry_pdq_iec,
             "siemens_flatness_difference": fa.flatness_dose_difference, "siemens_symmetry_area": fa.symmetry_area}
    METRICS = list(calcs)
    FIELD_KEYS = ["width (exact)", "beam center index (exact)", "beam center value (@rounded)", "cax index (exact)",
                  "cax value (@rounded)", "left index (exact)", "left value (@rounded)", "left slope",
                  "left intercept", "right slope", "right intercept", "left inner index (exact)",
                  "right inner index (exact)", '"top" index (exact)', '"top" value (@exact)', "right index (exact)",
                  "right value (@rounded)"]
    FWXM_KEYS = ["width (exact)", "center index (exact)", "center value (@rounded)", "left index (exact)",
                 "left value (@rounded)", "right index (exact)", "right value (@rounded)"]
    sg = {"metric_names": np.array(METRICS), "field_keys": np.array(FIELD_KEYS), "fwxm_keys": np.array(FWXM_KEYS)}

    def record(tag, p):
        fd = p.field_data(in_field_ratio=0.8, slope_exclusion_ratio=0.2)
        sg[f"{tag}.values"], sg[f"{tag}.x_indices"] = np.asarray(p.values, float), np.asarray(p.x_indices, float)
        sg[f"{tag}.field"] = np.array([float(fd[k]) for k in FIELD_KEYS])
        sg[f"{tag}.field_values"] = np.asarray(fd["field values"], float)
        sg[f"{tag}.top_params"] = np.asarray(fd["top params"], float)
        for hgt in (50, 25, 80):
            fw = p.fwxm_data(hgt)
            sg[f"{tag}.fwxm{hgt}"] = np.array([float(fw[k]) for k in FWXM_KEYS])
            sg[f"{tag}.fwxm{hgt}_values"] = np.asarray(fw["field values"], float)
        sg[f"{tag}.metrics"] = np.array([float(calcs[m](p, in_field_ratio=0.8)) for m in METRICS])

    modes = {"none": (prof.Interpolation.NONE, True, "expected_metrics"),
             "linear": (prof.Interpolation.LINEAR, True, "expected_metrics_linear"),
             "spline": (prof.Interpolation.SPLINE, True, "expected_metrics_spline"),
             "none_nox": (prof.Interpolation.NONE, False, "expected_metrics_no_x"),
             "linear_nox": (prof.Interpolation.LINEAR, False, "expected_metrics_linear_no_x"),
             "spline_nox": (prof.Interpolation.SPLINE, False, "expected_metrics_spline_no_x")}
    sg["n_fixtures"] = np.int64(len(fxm.PROFILE_REGRESSION_FIXTURES))
    for i, fx in enumerate(fxm.PROFILE_REGRESSION_FIXTURES):
        sg[f"fx{i}.x"], sg[f"fx{i}.y"] = np.asarray(fx.x_values, float), np.asarray(fx.values, float)
        # the reference's FROZEN expectations (tests_basic/core/test_profile.py:2546-2688; 1e-9 / 1e-4)
        sg[f"fx{i}.frozen_field_keys"] = np.array(list(fx.expected_field_data))
        sg[f"fx{i}.frozen_field"] = np.array([fx.expected_field_data[k] for k in fx.expected_field_data])
        for mode, (interp, use_x, attr) in modes.items():
            p = prof.SingleProfile(fx.values, x_values=fx.x_values if use_x else None, interpolation=interp)
            record(f"fx{i}.{mode}", p)
            frozen = getattr(fx, attr)
            sg[f"fx{i}.{mode}.frozen_metrics"] = np.array([frozen[m] for m in METRICS])
    # inflection-derivative edge method (row f4, first half) on the same fixtures
    INFL_KEYS = ["left index (exact)", "right index (exact)", "left value (@rounded)", "left value (@exact)",
                 "right value (@rounded)", "right value (@exact)"]
    sg["infl_keys"] = np.array(INFL_KEYS)
    for i, fx in enumerate(fxm.PROFILE_REGRESSION_FIXTURES):
        for mode, interp in (("none", prof.Interpolation.NONE), ("linear", prof.Interpolation.LINEAR)):
            try:
                p = prof.SingleProfile(fx.values, x_values=fx.x_values, interpolation=interp,
                                       edge_detection_method=prof.Edge.INFLECTION_DERIVATIVE)
                record(f"fx{i}.infl_{mode}", p)
                inf = p.inflection_data()
                sg[f"fx{i}.infl_{mode}.infl"] = np.array([float(inf[k]) for k in INFL_KEYS])
            except Exception as exc:   # a few coarse profiles have no gradient peak above 80 %: recorded as such
                sg[f"fx{i}.infl_{mode}.error"] = np.array(type(exc).__name__)
    # EPID-style profiles (pixel units, dpmm) through the options the fixtures do not touch
    epid = np.mean(synth_frames(1, 96, 400, seed=91)[0][40:56].astype(float), axis=0)
    opts = {"dpmm": dict(dpmm=1 / 0.336),
            "dpmm_spline": dict(dpmm=1 / 0.336, interpolation="Spline", interpolation_resolution_mm=0.05),
            "factor3": dict(interpolation_factor=3), "max": dict(normalization_method="Max"),
            "geo": dict(normalization_method="Geometric center", centering="Geometric center"),
            "raw": dict(normalization_method=None, ground=False, interpolation=None)}
    sg["epid.y"] = epid
    for name, kw in opts.items():
        record(f"epid.{name}", prof.SingleProfile(epid.copy(), **kw))
    np.savez_compressed(os.path.join(HERE, "single_profile.npz"), **sg)

    # ---- 11. field finder: the reference's own GlobalSizedFieldLocator under scikit-image 0.18.3 (a13, fields)
    ff, ff_dpmm, ff_args = field_frames(seed=61)
    ff = [f.astype(np.float32).astype(np.float64) for f in ff]
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "w.npz"), os.path.join(td, "f.npz")
        np.savez(inp, count=len(ff), dpmm=ff_dpmm, **{k: np.array([a[k] for a in ff_args], dtype=float) for k in ff_args[0]},
                 **{f"f{i}": f for i, f in enumerate(ff)})
        subprocess.run([PY39, os.path.join(HERE, "skimage_fields_py39.py"), inp, outp, ROOT], check=True)
        fg = dict(np.load(outp))
    for i, f in enumerate(ff):
        fg[f"{i}.frame"] = f.astype(np.float32)   # float32-representable by construction: exact, half the size
    fg["dpmm"] = np.float64(ff_dpmm)
    for k in ff_args[0]:
        fg[k] = np.array([a[k] for a in ff_args], dtype=float)
    np.savez_compressed(os.path.join(HERE, "fields.npz"), **fg)

    # ---- 12. DiskROI statistics: the reference's own class under scikit-image 0.18.3 (next row f3)
    rng = np.random.default_rng(71)
    ct_slice = catphan_slices(1, 512, seed=72)[0][0]
    rois = np.array([[256, 256, 10], [180.4, 300.7, 7.5], [330.5, 199.5, 12.25], [100, 120, 1], [256.2, 90.9, 30],
                     [400.75, 410.1, 45.5], [20.5, 20.5, 15]], dtype=float)
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "w.npz"), os.path.join(td, "f.npz")
        sl_f = (ct_slice.astype(np.float64) * 0.731 + rng.normal(0, 3, ct_slice.shape)).astype(np.float32).astype(np.float64)
        np.savez(inp, slice_i16=ct_slice, slice_f64=sl_f, rois=rois)
        subprocess.run([PY39, os.path.join(HERE, "skimage_roi_py39.py"), inp, outp, ROOT], check=True)
        rg = dict(np.load(outp))
    rg.update(slice_i16=ct_slice, slice_f32=sl_f.astype(np.float32), rois=rois)   # slice_f64 = slice_f32 exactly
    np.savez_compressed(os.path.join(HERE, "roi.npz"), **rg)

    # ---- 13. 2-D gamma: the reference's own gamma_2d (next row f4), its known-answer inputs + dose-like images
    rng = np.random.default_rng(81)
    cases = []

    def add(ref, ev, **kw):
        cases.append((np.asarray(ref, float), np.asarray(ev, float), kw))

    one = np.ones((5, 5))
    add(one, one)                                                     # test_gamma.py:108-121: all zeros
    add(one * 50, one * 50)
    add(one, one * 1.01, dose_to_agreement=1)                         # :122-136: exactly 1
    add(one, one * 0.99, dose_to_agreement=1)
    ev = one.copy(); ev[(0, 0, 1, 1), (0, 1, 1, 0)] = 1.03           # :137-170
    add(one, ev, dose_to_agreement=1, distance_to_agreement=1, gamma_cap_value=5)
    ref = one.copy(); ref[0, 0] = 100; ev = one.copy(); ev[0, 0] = 103; ev[0, 1] = 1.03   # :171-191 local dose
    add(ref, ev, dose_to_agreement=3, distance_to_agreement=1, gamma_cap_value=5, global_dose=False, dose_threshold=0)
    z = np.zeros((5, 5)); z[0, 0] = 1                                 # :193-230 threshold / fill value
    add(z, z, dose_to_agreement=3, distance_to_agreement=1, gamma_cap_value=5, global_dose=False, dose_threshold=5)
    add(z, z, dose_to_agreement=3, distance_to_agreement=1, gamma_cap_value=5, global_dose=False, dose_threshold=5,
        fill_value=0.666)
    add(one, one / 1.005, dose_to_agreement=1)                        # :232-238 half
    add(one, one * 10, dose_to_agreement=1, gamma_cap_value=2)        # :240-248 cap
    yy, xx = np.mgrid[0:72, 0:90]
    dose = 100 * np.exp(-(((yy - 35) / 22.0) ** 4 + ((xx - 44) / 28.0) ** 4))
    meas = 100 * np.exp(-(((yy - 35.6) / 22.3) ** 4 + ((xx - 43.2) / 27.7) ** 4)) * (1 + rng.normal(0, 0.01, dose.shape))
    for dta, dd, glob, thr in ((1, 1, True, 5), (2, 2, True, 10), (3, 3, False, 10), (4, 1, True, 0), (2, 3, False, 0)):
        add(dose, meas, dose_to_agreement=dd, distance_to_agreement=dta, global_dose=glob, dose_threshold=thr,
            gamma_cap_value=2 if dta < 3 else 1.5, fill_value=np.nan if dta != 2 else 0.0)
    hole = meas.copy(); hole[30:34, 40:47] = np.nan                  # NaNs in the evaluation are skipped by nanmin
    add(dose, hole, dose_to_agreement=2, distance_to_agreement=2)
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "w.npz"), os.path.join(td, "f.npz")
        pack = {"count": len(cases)}
        for k, (r_, e_, kw) in enumerate(cases):
            pack[f"ref{k}"], pack[f"ev{k}"], pack[f"kw{k}"] = r_, e_, np.array(kw, dtype=object)
        np.savez(inp, **pack)
        subprocess.run([PY39, os.path.join(HERE, "skimage_gamma_py39.py"), inp, outp, ROOT], check=True)
        gg = dict(np.load(outp))
    gg["count"] = np.int64(len(cases))
    for k, (r_, e_, kw) in enumerate(cases):
        gg[f"ref{k}"], gg[f"ev{k}"] = r_, e_
        gg[f"kw{k}"] = np.array(json.dumps({a: (None if isinstance(b, float) and np.isnan(b) else b) for a, b in kw.items()}))
    np.savez_compressed(os.path.join(HERE, "gamma.npz"), **gg)

    # ---- 14. 1-D gamma: the reference's own gamma_1d (py3.10: only scipy's interp1d underneath)
    gmod = ref_loader.ref("core.gamma")
    rng = np.random.default_rng(91)
    xr = np.linspace(-60, 60, 241)
    prof_r = 100 / (1 + np.exp((np.abs(xr) - 40) / 2.5)) + 0.5
    xe = np.sort(np.concatenate([np.linspace(-61, 61, 170), rng.uniform(-50, 50, 30)]))
    prof_e = (100 / (1 + np.exp((np.abs(xe - 0.4) - 40.3) / 2.6)) + 0.5) * (1 + rng.normal(0, 0.008, xe.size))
    g1 = {}
    g1_cases = [
        dict(reference=np.ones(5), evaluation=np.ones(5)),                                     # test_gamma.py:304-316
        dict(reference=np.ones(5), evaluation=np.ones(5) * 1.01, dose_to_agreement=1),          # :318-331
        dict(reference=np.ones(5), evaluation=np.ones(5) / 1.005, dose_to_agreement=1),         # :333-339
        dict(reference=np.ones(5), evaluation=np.array([1.03, 1.03, 1, 1, 1]), dose_to_agreement=1,
             distance_to_agreement=1, gamma_cap_value=5),                                       # :341-367
        dict(reference=np.array([100, 1, 1, 1, 1.0]), evaluation=np.array([103, 1.03, 1, 1, 1]), dose_to_agreement=3,
             gamma_cap_value=5, global_dose=False, dose_threshold=0),                           # :369-380
        dict(reference=np.array([1, 0, 0, 0, 0.0]), evaluation=np.array([1, 0, 0, 0, 0.0]), dose_to_agreement=3,
             gamma_cap_value=5, global_dose=False, dose_threshold=5, fill_value=0.666),         # :382-415
        dict(reference=np.ones(5), evaluation=np.ones(5) * 10, dose_to_agreement=1, gamma_cap_value=2),   # :417-425
        dict(reference=prof_r, evaluation=prof_e, reference_coordinates=xr, evaluation_coordinates=xe,
             dose_to_agreement=1, distance_to_agreement=1, resolution_factor=5),
        dict(reference=prof_r, evaluation=prof_e, reference_coordinates=xr, evaluation_coordinates=xe,
             dose_to_agreement=2, distance_to_agreement=2, global_dose=False, dose_threshold=10, fill_value=0.0),
        dict(reference=prof_r, evaluation=prof_e[::-1].copy(), reference_coordinates=xr,
             evaluation_coordinates=xe[::-1].copy(), dose_to_agreement=3, distance_to_agreement=0.5,
             resolution_factor=8, gamma_cap_value=1.2),                                         # reversed abscissae
    ]
    g1["count"] = np.int64(len(g1_cases))
    for k, kw in enumerate(g1_cases):
        gam, vals, xs_ = gmod.gamma_1d(**kw)
        g1[f"gamma{k}"], g1[f"vals{k}"], g1[f"xs{k}"] = gam, vals, xs_
        for name in ("reference", "evaluation", "reference_coordinates", "evaluation_coordinates"):
            if name in kw:
                g1[f"{name}{k}"] = np.asarray(kw[name], float)
        g1[f"kw{k}"] = np.array(json.dumps({a: b for a, b in kw.items() if not isinstance(b, np.ndarray)}))
    np.savez_compressed(os.path.join(HERE, "gamma1d.npz"), **g1)

    # ---- 15. XIM files: synthetic compressed .xim files read by the reference's own XIM reader (next row f1)
    image_mod = ref_loader.ref("core.image")
    from oracle import pylinac_oracle as orc

    rng = np.random.default_rng(101)
    xg = {}

    def blob(h, w, noise, spikes):
        yy, xx = np.mgrid[0:h, 0:w]
        img = 20000 + 15000 * np.exp(-(((yy - h / 2) / (h / 3)) ** 2 + ((xx - w / 2) / (w / 3)) ** 2)) + rng.normal(0, noise, (h, w))
        img = img.round().astype(np.int64)
        img.ravel()[rng.integers(0, h * w, spikes)] = rng.choice([0, 65535, 1 << 20, -(1 << 18)], spikes)
        return img

    xim_cases = {"a": (blob(260, 300, 40, 6), 4), "b": (blob(150, 200, 300, 4), 2), "c": (blob(64, 64, 40000, 30), 4),
                 "d": (blob(2, 5, 10, 0), 4)}   # (a one-row image has an empty lookup table: the reference raises IndexError)
    for name, (img, bpp) in xim_cases.items():
        props = {"PixelWidth": 0.0336, "PixelHeight": 0.0336, "MVBeamOn": 1, "AcquisitionSystemVersion": "3.1.2",
                 "KVCollimatorShape": np.array([1.5, 2.5, -3.0]), "Couch": np.array([3, 4, 5])}
        data = orc.xim_file_bytes(img, bpp, props, histogram=tuple(range(7)))
        with tempfile.NamedTemporaryFile(suffix=".xim", delete=False) as f:
            f.write(data)
            path = f.name
        x = image_mod.XIM(path)
        os.unlink(path)
        xg[f"{name}.file"] = np.frombuffer(data, dtype=np.uint8)
        xg[f"{name}.array"] = x.array
        xg[f"{name}.dpmm"] = np.float64(x.dpmm)
        xg[f"{name}.histogram"] = np.asarray(x.histogram)
        xg[f"{name}.props"] = np.array(json.dumps({k: (np.asarray(v).tolist() if not isinstance(v, (str, int, float)) else v)
                                                   for k, v in x.properties.items()}))
    np.savez_compressed(os.path.join(HERE, "xim.npz"), **xg)

    # ---- 16. Canny: scikit-image 0.18.3's own feature.canny with pylinac's parameters (next row f2)
    from scipy import ndimage as ndi

    rng = np.random.default_rng(111)

    def planar_phantom(h, w, noise):
        """kV-phantom-like frame: rotated square body on a gradient background, inner discs, blur, noise; float64."""
        yy, xx = np.mgrid[0:h, 0:w].astype(float)
        a = np.deg2rad(7.0)
        u = (xx - w / 2) * np.cos(a) + (yy - h / 2) * np.sin(a)
        v = -(xx - w / 2) * np.sin(a) + (yy - h / 2) * np.cos(a)
        img = 0.25 + 0.1 * xx / w
        img[(np.abs(u) < w * 0.28) & (np.abs(v) < h * 0.3)] = 0.7
        for k, (du, dv) in enumerate([(-0.12, -0.1), (0.1, -0.12), (0.0, 0.1), (0.15, 0.12)]):
            img[np.hypot(u - du * w, v - dv * h) < 9 + 2 * k] = 0.45 + 0.1 * k
        return ndi.gaussian_filter(img, 1.2) + rng.normal(0, noise, img.shape)

    cimgs = [planar_phantom(200, 260, 0.004), planar_phantom(200, 260, 0.004) * 4000.0, planar_phantom(151, 97, 0.01),
             np.pad(np.ones((40, 40)), 30) + 0.2 * rng.random((100, 100)), np.zeros((32, 48))]
    ckw = [dict(sigma=2, low_threshold=0.001, high_threshold=0.01, use_quantiles=True),          # planar_imaging.py:198
           dict(sigma=4, low_threshold=0.001, high_threshold=0.01, use_quantiles=True),          # :1978
           dict(sigma=1.5, low_threshold=0.5, high_threshold=0.9, use_quantiles=True),
           dict(sigma=1.0), dict(sigma=2.0, low_threshold=0.02, high_threshold=0.05)]
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "w.npz"), os.path.join(td, "f.npz")
        pack = {"count": len(cimgs)}
        for k, (im, kw) in enumerate(zip(cimgs, ckw)):
            pack[f"img{k}"], pack[f"kw{k}"] = im, np.array(kw, dtype=object)
        np.savez(inp, **pack)
        subprocess.run([PY39, os.path.join(HERE, "skimage_canny_py39.py"), inp, outp, ROOT], check=True)
        cg = dict(np.load(outp))
    cg["count"] = np.int64(len(cimgs))
    for k, (im, kw) in enumerate(zip(cimgs, ckw)):
        cg[f"img{k}"], cg[f"kw{k}"] = im, np.array(json.dumps(kw))
    np.savez_compressed(os.path.join(HERE, "canny.npz"), **cg)

    # ---- 17. hough_line: scikit-image 0.18.3's own compiled transform.hough_line (next row f2, second half)
    subprocess.run([PY39, os.path.join(HERE, "skimage_hough_py39.py"), os.path.join(HERE, "hough.npz")], check=True)

    # ---- 18. BaseImage.gamma (Bakai map, a15): the reference's own ArrayImage.gamma (its unit test is @skip)
    rng = np.random.default_rng(121)
    fr = synth_frames(2, 160, 200, seed=122).astype(np.float64)
    ref_u16 = fr[0].astype(np.uint16)
    cmp_u16 = np.clip(np.roll(fr[0], (1, -2), (0, 1)) * 1.01 + rng.normal(0, 150, fr[0].shape), 0, 65535).astype(np.uint16)
    bg = {}
    bk_cases = {"u16": (ref_u16, cmp_u16, dict()), "u16_opts": (ref_u16, cmp_u16, dict(doseTA=2, distTA=3, threshold=0.3)),
                "f64": (ref_u16 / 65535.0, cmp_u16 / 65535.0, dict(doseTA=3, distTA=1, ground=False)),
                "f64_raw": (fr[1] / 1000.0, fr[1] / 1000.0 * (1 + rng.normal(0, 0.01, fr[1].shape)),
                            dict(doseTA=1, distTA=2, ground=False, normalize=False, threshold=0.05))}
    for name, (a, b, kw) in bk_cases.items():
        ia, ib = image.ArrayImage(a.copy(), dpi=75.6), image.ArrayImage(b.copy(), dpi=75.6)
        bg[f"{name}.ref"], bg[f"{name}.cmp"] = a, b
        bg[f"{name}.kw"] = np.array(json.dumps(kw))
        bg[f"{name}.gamma"] = ia.gamma(ib, **kw)
    np.savez_compressed(os.path.join(HERE, "bakai.npz"), **bg)

    # ---- 19. planar phantom outline + hough_line_peaks: scikit-image 0.18.3's own canny / label / regionprops /
    #          hough_line / hough_line_peaks on synthetic phantom frames (next row f2: planar_imaging.py:300-341, 3136-3179)
    subprocess.run([PY39, os.path.join(HERE, "skimage_planar_py39.py"), os.path.join(HERE, "planar.npz")], check=True)

    # ---- 20. RectangleROI: the reference's own class (rotated and unrotated) + raw skimage.draw.polygon pixel lists
    #          (next row f3, second half); reads the slices of roi.npz
    subprocess.run([PY39, os.path.join(HERE, "skimage_rect_py39.py"), os.path.join(HERE, "roi.npz"),
                    os.path.join(HERE, "rect.npz"), ROOT], check=True)

    # ---- 21. contrast ROIs: the reference's own LowContrastDiskROI / HighContrastDiskROI + core.contrast (f3)
    subprocess.run([PY39, os.path.join(HERE, "skimage_contrast_py39.py"), os.path.join(HERE, "roi.npz"),
                    os.path.join(HERE, "contrast.npz"), ROOT], check=True)
    # ---- 22. canny on integer images (scikit-image 0.18.3 itself)
    subprocess.run([PY39, os.path.join(HERE, "skimage_canny_int_py39.py"), os.path.join(HERE, "canny_int.npz")], check=True)
    subprocess.run([PY39, os.path.join(HERE, "skimage_canny_mask_py39.py"), os.path.join(HERE, "canny_mask.npz")], check=True)
    # ---- 23. ThicknessROI: the reference's own pylinac.ct.ThicknessROI on synthetic wire ramps
    subprocess.run([PY39, os.path.join(HERE, "skimage_thickness_py39.py"), os.path.join(HERE, "thickness.npz"), ROOT], check=True)
    # ---- 24. CatPhan volume localisation: the reference's own find_phantom_axis / find_origin_slice
    subprocess.run([PY39, os.path.join(HERE, "skimage_catphan_volume_py39.py"), os.path.join(HERE, "catphan_volume.npz"), ROOT],
                   check=True)
    # (hill.npz and starshot.npz have their own generators: make_hill_golden.py, make_starshot_golden.py)

    json.dump(meta, open(os.path.join(HERE, "META.json"), "w"), indent=1)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    if not ref_loader.reference_available():
        raise SystemExit("needs /root/reference (build container only)")
    main()
