"""Parity checks of the "next"-row host functions that are shared by the `-m gpu` tests (real MI355X) and the
emulated-device tests (tests/emu_backend.py, CPU).  Each takes the golden loader and the device."""
from __future__ import annotations

import numpy as np
import pytest
import torch


def _peaks_kwargs(row):
    md, ma, thr, npk = row
    kw = dict(min_distance=int(md), min_angle=int(ma))
    if thr >= 0:
        kw["threshold"] = float(thr)
    if npk >= 0:
        kw["num_peaks"] = int(npk)
    return kw


def check_hough_line_peaks(golden, dev):
    """planar.hough_line_peaks == scikit-image 0.18.3 transform.hough_line_peaks (tests/golden/planar.npz)."""
    from pylinac_amd import planar

    g = golden("planar")
    for k in range(4):
        hs, an, di = g[f"acc{k}.hspace"], g[f"acc{k}.angles"], g[f"acc{k}.dists"]
        for j in range(4):
            h, a, d = planar.hough_line_peaks(torch.from_numpy(hs.astype(np.int64)).to(dev), an, di,
                                              **_peaks_kwargs(g[f"acc{k}.kw{j}"]))
            assert np.array_equal(h, g[f"acc{k}.p{j}.h"]), (k, j)
            if (k, j) == (2, 2):
                # the top-`num_peaks` cut falls inside a run of equal heights: np.argsort (unstable) decides, and its
                # tie order differs between numpy versions -- the reference's choice is not defined there
                continue
            assert np.array_equal(a, g[f"acc{k}.p{j}.a"]) and np.array_equal(d, g[f"acc{k}.p{j}.d"]), (k, j)
    for n in g["names"]:
        hs, an, di = g[f"{n}.hspace"], g[f"{n}.theta"], g[f"{n}.dists"]
        for md in (17, 9):
            for npk, tag in ((2, "2"), (np.inf, "inf")):
                h, a, d = planar.hough_line_peaks(hs, an, di, min_distance=md, num_peaks=npk, device=dev)
                t = f"{n}.peaks.md{md}.n{tag}"
                assert np.array_equal(h, g[t + ".h"]) and np.array_equal(a, g[t + ".a"]) and \
                    np.array_equal(d, g[t + ".d"]), t


def check_phantom_outline(golden, dev, names=None):
    """canny -> label -> bbox table -> phantom_ski_region -> region.image -> hough_line, against scikit-image's own
    regionprops / hough_line on the same frames."""
    from pylinac_amd import canny, planar

    g = golden("planar")
    for n in (names or g["names"]):
        img, (sigma, lo, hi) = g[f"{n}.img"], g[f"{n}.kw"]
        edges, labels, tables = planar.canny_regions(torch.from_numpy(img).to(dev), sigma=sigma, percentiles=(lo, hi))
        assert np.array_equal(edges[0].cpu().numpy().astype(bool), g[f"{n}.edges"]), n
        assert np.array_equal(tables[0], g[f"{n}.bbox"]), n
        big = int(g[f"{n}.big"])
        size = float(g[f"{n}.bbox_area"][big])
        region = planar.find_phantom_region(torch.from_numpy(img).to(dev), size, sigma=sigma, percentiles=(lo, hi))
        assert region.label == big + 1 and region.bbox == tuple(int(v) for v in g[f"{n}.bbox"][big]), n
        assert region.bbox_area == size
        assert np.array_equal(region.image.cpu().numpy(), g[f"{n}.region_image"]), n
        assert np.allclose(region.centroid, g[f"{n}.centroid"], rtol=0, atol=1e-9), n
        assert abs(region.orientation - float(g[f"{n}.orientation"])) < 1e-9, (n, region.orientation)
        hs, _, di = canny.hough_line(region.image, theta=g[f"{n}.theta"])
        assert np.array_equal(hs.cpu().numpy().astype(np.uint64), g[f"{n}.hspace"]), n
        assert np.array_equal(di, g[f"{n}.dists"]), n


# regions whose second moments are symmetric (a - c == 0 analytically) AND whose scikit-image value came out of the
# general atan2 branch because np.dot's rounding left a - c = 1e-15: scikit-image's special-case branch has the opposite
# sign convention to the limit of its general branch (b < 0 -> -pi/4 vs atan2(-2b, 0+) / 2 = +pi/4), so on such regions the
# reference's sign is BLAS-rounding noise.  This build takes the special-case branch there (exact a - c == 0).
REGIONPROPS_SIGN_IS_NOISE = ("blob_sym",)


def raw_moments_numpy(labels, label=1):
    r, c = np.nonzero(labels == label)
    r, c = r.astype(object), c.astype(object)
    return (len(r), int(r.sum()), int(c.sum()), int((r * r).sum()), int((c * c).sum()), int((r * c).sum()))


def check_regionprops_formulas(g, raw_of):
    """centroid / orientation / eccentricity / inertia tensor formed from exact raw moments against scikit-image 0.18.3's
    regionprops (tests/golden/regionprops.npz); ``raw_of(labels) -> (m00, m10, m01, m20, m02, m11)``."""
    from pylinac_amd import regionprops as rp

    for n in g["names"]:
        raw = raw_of(g[f"{n}.labels"])
        assert raw == raw_moments_numpy(g[f"{n}.labels"]), n
        assert np.allclose(rp.centroid(raw), g[f"{n}.centroid"], rtol=0, atol=1e-12), n
        assert np.allclose(rp.inertia_tensor(raw), g[f"{n}.inertia_tensor"], rtol=1e-12, atol=1e-12), n
        # sqrt(1 - l2/l1) near 0 amplifies the eigenvalues' rounding: absolute 1e-6 there, 1e-12 elsewhere
        assert abs(rp.eccentricity(raw) - float(g[f"{n}.eccentricity"])) < 1e-6, n
        want = float(g[f"{n}.orientation"])
        if n in REGIONPROPS_SIGN_IS_NOISE:
            assert abs(abs(rp.orientation(raw)) - np.pi / 4) < 1e-12 and abs(abs(want) - np.pi / 4) < 1e-12, n
        else:
            assert abs(rp.orientation(raw) - want) < 1e-12, (n, rp.orientation(raw), want)


def check_region_moments_kernel(golden, dev):
    """pl_region_moments on the golden label images (one label each) + a multi-label, multi-frame batch against numpy"""
    from pylinac_amd import ops

    g = golden("regionprops")

    def raw_of(labels):
        mom, ovf = ops.region_moments(torch.from_numpy(labels).to(dev), 1)
        assert int(ovf.max()) == 0
        return tuple(int(v) for v in mom[0, 0].cpu().tolist())

    check_regionprops_formulas(g, raw_of)
    rng = np.random.default_rng(3)
    lab = rng.integers(0, 6, (3, 37, 131)).astype(np.int32)        # rows shorter than / straddling a 64-lane wave
    lab[1] = 4                                                      # a frame that is one region
    mom, ovf = ops.region_moments(torch.from_numpy(lab).to(dev), 5)
    mom = mom.cpu().numpy()
    assert int(ovf.max()) == 0
    for f in range(3):
        for k in range(1, 6):
            assert tuple(int(v) for v in mom[f, k - 1]) == raw_moments_numpy(lab[f], k), (f, k)
    _, ovf = ops.region_moments(torch.from_numpy(lab).to(dev), 3)
    assert ovf.cpu().tolist() == [1, 1, 1]


def check_otsu16(golden, dev, big=True):
    """pl_otsu16 (single-pass LDS-window Otsu; window placed by a row sample or by the caller's bounds; gated two-kernel
    path for frames that do not fit) against the golden skimage 0.18.3 thresholds and the oracle's restatement."""
    from oracle import pylinac_oracle as o
    from pylinac_amd import ops

    g = golden("otsu")
    for k in ["u16_field", "u16_random", "i16", "const", "two_level"]:
        a = g[f"{k}.in"]
        t = torch.from_numpy(a).to(dev)
        x = t if t.ndim == 3 else t[None]
        flat = x.cpu().numpy().reshape(x.shape[0], -1).astype(np.int64)
        lo, hi = flat.min(1), flat.max(1)
        thr, mn, mx = ops.otsu16(x)
        assert np.array_equal(thr.cpu().numpy(), np.atleast_1d(g[f"{k}.otsu"])), k
        assert np.array_equal(mn.cpu().numpy(), lo) and np.array_equal(mx.cpu().numpy(), hi), k
        # caller's (enclosing) bounds give the same answer
        info = np.iinfo(a.dtype)
        tlo = torch.from_numpy(np.clip(lo - 7, info.min, info.max).astype(np.int32)).to(dev)
        thi = torch.from_numpy(np.clip(hi + 11, info.min, info.max).astype(np.int32)).to(dev)
        thr2, mn2, mx2 = ops.otsu16(x, tlo, thi)
        assert np.array_equal(thr2.cpu().numpy(), thr.cpu().numpy()) and np.array_equal(mn2.cpu().numpy(), lo) and \
            np.array_equal(mx2.cpu().numpy(), hi), k
    rng = np.random.default_rng(91)
    shapes = [(3, 40, 64), (2, 33, 50), (1, 21, 37)] + ([(3, 160, 1024)] if big else [])
    for shape in shapes:
        for dt in (np.uint16, np.int16):
            a = (rng.integers(2000, 2400, shape) + (np.arange(shape[2]) > shape[2] // 2) * 9000).astype(np.int64)
            a[-1] = rng.integers(0, 65536, shape[1:])            # last frame: full range -> two-kernel path
            if shape[0] > 2:                                      # a frame whose extrema sit in ONE unsampled pixel each
                a[1, shape[1] - 1, 3] = 60000
                a[1, 1, 5] = 17
            a = (a - (32768 if dt == np.int16 else 0)).astype(dt)
            thr, tmin, tmax = ops.otsu16(torch.from_numpy(a).to(dev))
            assert np.array_equal(thr.cpu().numpy(), np.array([o.threshold_otsu(f) for f in a])), (shape, dt)
            assert np.array_equal(tmin.cpu().numpy(), a.reshape(shape[0], -1).min(1)), (shape, dt)
            assert np.array_equal(tmax.cpu().numpy(), a.reshape(shape[0], -1).max(1)), (shape, dt)
    # flat frames (one value per whole wave: the single-atomic path of both histogram kernels), ragged sizes so that the
    # last vectors are handled by partial waves, a few outliers in otherwise constant data
    for shape in ((2, 63, 72), (1, 200, 1024), (1, 37, 50)) if big else ((2, 63, 72), (1, 37, 50)):
        a = np.zeros(shape, dtype=np.uint16)
        a[0, shape[1] // 2:, :] = 31000
        a[0, 3, 5] = 7
        a[-1, -1, -1] = 40000
        t = torch.from_numpy(a).to(dev)
        h = ops.histogram16(t).cpu().numpy().astype(np.int64)
        assert np.array_equal(h, np.stack([np.bincount(f.ravel(), minlength=65536) for f in a])), shape
        thr, tmin, tmax = ops.otsu16(t)
        assert np.array_equal(thr.cpu().numpy(), np.array([o.threshold_otsu(f) for f in a])), shape
        assert np.array_equal(tmin.cpu().numpy(), a.reshape(shape[0], -1).min(1)) and \
            np.array_equal(tmax.cpu().numpy(), a.reshape(shape[0], -1).max(1)), shape


def check_wl_analyze_batch(golden, dev, frames=None):
    """winston_lutz.analyze_batch against the reference's own per-image sequence (check_inversion_by_histogram ->
    _clean_edges -> ground -> normalize -> find_field_centroids -> find_bb_centroids, driven under scikit-image 0.18.3;
    tests/golden/skimage_wl_py39.py): config #4 recipe frames incl. an inverted one, one whose edges get cropped
    twice and an off-centre one.  Field CAX exact (integer sums / count), BB weighted centroid 1e-9."""
    from pylinac_amd import winston_lutz as wl

    g = golden("wl")
    sel = list(range(len(g["frames"]))) if frames is None else list(frames)
    fr = torch.from_numpy(g["frames"][sel]).to(dev)
    res = wl.analyze_batch(fr, 1 / float(g["pixel_mm"]), float(g["bb_mm"]))
    want = g["record"][sel]
    assert np.array_equal(res["status"], np.zeros(len(sel), dtype=np.int32))
    assert np.array_equal(res["inverted"], g["inverted"][sel])
    assert np.array_equal(res["crop"] * 2, g["frames"].shape[1] - g["shape_after_clean"][sel][:, 0])
    assert np.array_equal(res["record"][:, :2], want[:, :2]), np.abs(res["record"][:, :2] - want[:, :2]).max()
    assert np.allclose(res["record"][:, 2:], want[:, 2:4], rtol=0, atol=1e-9), np.abs(res["record"][:, 2:] - want[:, 2:4]).max()
    if frames is None:
        # a batch that keeps every frame AND holds an inverted one (no edge-cleaned frame: the whole-batch branch)
        sub = [0, 1, 6, 9]
        r2 = wl.analyze_batch(torch.from_numpy(g["frames"][sub]).to(dev), 1 / float(g["pixel_mm"]), float(g["bb_mm"]))
        assert np.array_equal(r2["record"], res["record"][sub]) and np.array_equal(r2["inverted"], g["inverted"][sub])


def check_wl_analyze_batch_other_dtypes(golden, dev, frames=(0, 6, 7), int16_frames=None):
    """winston_lutz.analyze_batch on what the reference's loader produces -- float64 frames (stored value
    * RescaleSlope + RescaleIntercept, pydicom's apply_rescale) and int16 frames -- against the oracle's restatement of the
    per-image sequence run on exactly those arrays (field CAX exact, BB 1e-9), and against the uint16 result of the same
    pixels (an affine map with a positive slope moves nothing: same CAX, BB within 1e-9); plus the refused cases."""
    from oracle import pylinac_oracle as orc
    from pylinac_amd import winston_lutz as wl

    g = golden("wl")
    sel = list(frames)
    dpmm, bb_mm = 1 / float(g["pixel_mm"]), float(g["bb_mm"])
    u16 = g["frames"][sel]
    base = wl.analyze_batch(torch.from_numpy(u16).to(dev), dpmm, bb_mm)
    scaled = u16.astype(np.float64) * 4.315e-5
    scaled += -0.25
    i16 = (u16.astype(np.int32) // 2 - 16384).astype(np.int16)                     # range < 32768: the reference's int16 path is exact
    keep16 = list(range(len(sel))) if int16_frames is None else [sel.index(f) for f in int16_frames]
    for arr in (scaled, i16):
        if arr.dtype == np.int16 and len(keep16) < len(sel):     # (the emulated suite runs the int16 form on fewer frames)
            arr, base = arr[keep16], {k: v[keep16] for k, v in base.items()}
        res = wl.analyze_batch(torch.from_numpy(arr).to(dev), dpmm, bb_mm)
        for k, f in enumerate(arr):
            fx, fy, bx, by, inv, crop = orc.wl_analyze_frame(f, dpmm, bb_mm)
            assert res["record"][k, 0] == fx and res["record"][k, 1] == fy, (arr.dtype, k)
            assert np.allclose(res["record"][k, 2:], [bx, by], rtol=0, atol=1e-9), (arr.dtype, k)
            assert bool(res["inverted"][k]) == bool(inv) and int(res["crop"][k]) == int(crop), (arr.dtype, k)
        assert np.array_equal(res["status"], base["status"]) and np.array_equal(res["crop"], base["crop"])
        if arr.dtype == np.float64:
            assert np.array_equal(res["record"][:, :2], base["record"][:, :2])
            assert np.allclose(res["record"][:, 2:], base["record"][:, 2:], rtol=0, atol=1e-9)
    wide = np.zeros((1, 64, 64), np.int16)
    wide[0, :8] = -30000
    wide[0, 8:] = 30000
    with pytest.raises(ValueError):
        wl.analyze_batch(torch.from_numpy(wide).to(dev), dpmm, bb_mm)
    with pytest.raises(TypeError):
        wl.analyze_batch(torch.from_numpy(u16.astype(np.float32)).to(dev), dpmm, bb_mm)


def check_field_cax(dev):
    """ops.field_cax (pl_field_cax: one streaming reduction + an LDS window flood fill) against scipy's
    binary_fill_holes + center_of_mass on masks with holes, nested holes, shapes touching the frame border (a hole that
    is open to the border is not a hole), several blobs, an empty mask, and a foreground too large for the LDS window
    (general path).  Exact: integer sums / count."""
    from scipy import ndimage

    from pylinac_amd import ops

    rng = np.random.default_rng(12)
    frames = []
    h, w = 96, 130
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.zeros((h, w), bool); a[30:60, 40:90] = True; a[40:50, 55:70] = False; a[43:47, 60:64] = True   # hole with an island
    frames.append(a)
    b = np.zeros((h, w), bool); b[0:40, 0:50] = True; b[10:20, 0:12] = False; b[25:30, 20:30] = False      # open to the border / closed
    frames.append(b)
    c = (np.hypot(yy - 50, xx - 60) < 30) & ~(np.hypot(yy - 50, xx - 60) < 12); c |= np.hypot(yy - 80, xx - 115) < 9
    frames.append(c)
    frames.append(np.zeros((h, w), bool))                                                                  # empty
    d = rng.random((h, w)) < 0.55; frames.append(d)                                                        # noise: many holes
    e = np.ones((h, w), bool); e[1:-1, 1:-1] = False; e[40:50, 40:50] = True; frames.append(e)             # a ring along the border
    f = np.zeros((h, w), bool); f[:, 64] = True; f[48, :] = True; frames.append(f)                         # a cross: no holes
    m = np.stack(frames)
    x = torch.from_numpy((m * 1000 + 7).astype(np.uint16)).to(dev)
    got = ops.field_cax(x, 7.0, 1000.0, 0.5).cpu().numpy()
    for k, fm in enumerate(m):
        filled = ndimage.binary_fill_holes(fm)
        if not filled.any():
            assert np.isnan(got[k, 0]) and np.isnan(got[k, 1]) and got[k, 2] == 0, k
            continue
        want = ndimage.center_of_mass(filled)
        assert got[k, 2] == filled.sum(), (k, got[k, 2], filled.sum())
        assert np.array_equal(got[k, :2], np.array(want)), (k, got[k], want)
    big = np.zeros((2, 420, 440), bool); big[:, 10:410, 15:430] = True; big[0, 100:200, 100:300] = False; big[1, 0:50, 200:210] = False
    xb = torch.from_numpy((big * 500).astype(np.uint16)).to(dev)
    gb = ops.field_cax(xb, 0.0, 500.0, 0.5).cpu().numpy()
    for k in range(2):
        filled = ndimage.binary_fill_holes(big[k])
        assert gb[k, 2] == filled.sum() and np.array_equal(gb[k, :2], np.array(ndimage.center_of_mass(filled))), k


def check_ctp528_batch(golden, dev, whole=True):
    """ct.ctp528_batch (combine +-3 slices by max -> collapsed circle profile -> 1-D Gaussian -> ground -> per-region
    peaks / valleys -> relative MTF) against the reference's OWN CTP528CP504.circle_profile / .mtf per slice on a
    synthetic CatPhan volume (tests/golden/skimage_ctp528_py39.py), incl. the phantom-axis fits from the reference's
    find_phantom_axis.  Profiles 1e-9, rMTF 1e-9, same number of regions."""
    from pylinac_amd import ct

    g = golden("ctp528")
    vol = torch.from_numpy(g["volume"]).to(dev)
    mmpp = float(g["mmpp"])
    sl = g["slices"]
    prof, idx = ct.ctp528_profiles_batch(vol, mmpp, g["fit_zx"], g["fit_zy"], slices=sl)
    assert np.array_equal(idx, sl)
    assert np.allclose(prof.cpu().numpy(), g["profiles"], rtol=0, atol=1e-9)
    res = ct.ctp528_mtf_batch(prof)
    assert np.array_equal(res["nregions"], g["nregions"])
    for key in ("rmtf", "maxs", "mins"):
        assert np.allclose(res[key], g[key], rtol=1e-9, atol=1e-9, equal_nan=True), key
    # a profile without line pairs: the reference raises "Did not find any spatial resolution pairs" -> all NaN, 0 regions
    flat = torch.zeros((1, prof.shape[1]), dtype=torch.float64, device=prof.device)
    r0 = ct.ctp528_mtf_batch(flat)
    assert int(r0["nregions"][0]) == 0 and np.isnan(r0["rmtf"]).all()
    if whole:   # the composed entry: axis fits from the device's own phantom ROIs, every slice of the volume
        full = ct.ctp528_batch(vol, mmpp)
        assert np.allclose(full["fit_zx"], g["fit_zx"], rtol=1e-9, atol=1e-9) and np.allclose(full["fit_zy"], g["fit_zy"], rtol=1e-9, atol=1e-9)
        assert np.allclose(full["rmtf"][sl], g["rmtf"], rtol=1e-7, atol=1e-7, equal_nan=True)
        # several volumes in one batch never mix: [volume, volume shifted by 5 px] == the two single-volume results
        v2 = torch.roll(vol, shifts=(5, -3), dims=(1, 2))
        both = ct.ctp528_batch(torch.stack([vol, v2]), mmpp)
        one = ct.ctp528_batch(v2, mmpp)
        s_ = len(vol)
        assert np.array_equal(both["rmtf"][:s_], full["rmtf"], equal_nan=True) and np.array_equal(both["rmtf"][s_:], one["rmtf"], equal_nan=True)
        assert np.array_equal(both["fit_zx"][0], full["fit_zx"]) and np.array_equal(both["fit_zx"][1], one["fit_zx"])
        # (the last three slices of every volume hold NaN profiles: their +-3 window passes the end of the stack)
        assert torch.equal(torch.nan_to_num(both["profiles"][s_:], nan=-1.0), torch.nan_to_num(one["profiles"], nan=-1.0))
        assert bool(torch.isnan(one["profiles"][-3:]).all()) and not bool(torch.isnan(one["profiles"][:-3]).any())
        # one volume per chunk, slices confined to the LAST volume: chunks without a requested slice still
        # contribute their fits and ROI rows, and the centres come from the right volume's fit
        pick = np.array([s_ + int(sl[0]), s_ + int(sl[-1])])
        late = ct.ctp528_batch(torch.stack([vol, v2]), mmpp, slices=pick, chunk_volumes=1)
        assert np.array_equal(late["slices"], pick) and late["fit_zx"].shape == (2, 2) and len(late["roi"]) == 2 * s_
        assert np.array_equal(late["fit_zx"], both["fit_zx"]) and np.array_equal(late["fit_zy"], both["fit_zy"])
        assert np.array_equal(late["rmtf"], both["rmtf"][pick], equal_nan=True)
        assert np.array_equal(late["center"], both["center"][pick])


def check_rectangle_roi(golden, dev):
    """RectangleROI / polygon statistics against the reference's own RectangleROI and raw skimage.draw.polygon pixel
    lists (tests/golden/rect.npz): counts, min, max, median exact; mean / std to 1e-12 (summation order)."""
    from pylinac_amd import roi

    g, d = golden("rect"), golden("roi")
    for name, arr in (("i16", d["slice_i16"]), ("f32", d["slice_f32"])):
        frames = torch.from_numpy(arr).to(dev)[None]
        out, status = roi.rectangle_roi_stats_batch(frames, g["rects"])
        out, status = out.cpu().numpy()[0], status.cpu().numpy()[0]
        assert (status == 0).all()
        want = g["stats_" + name]
        assert np.array_equal(out[:, [0, 3, 4, 5]], want[:, [0, 3, 4, 5]]), name
        np.testing.assert_allclose(out[:, 1:3], want[:, 1:3], rtol=1e-12, atol=1e-12)
    # the class, with the reference's attribute names
    k = 2
    w, h, cx, cy, rot = g["rects"][k]
    m = roi.RectangleROI(d["slice_i16"], width=w, height=h, center=(cx, cy), rotation=rot)
    assert np.allclose(m.vertices, g["vertices"][k], rtol=0, atol=1e-12)
    want = g["stats_i16"][k]
    assert m.min == want[3] and m.max == want[4] and abs(m.mean - want[1]) < 1e-10 and abs(m.std - want[2]) < 1e-10
    assert m.pixel_value == m.mean and m.area == w * h
    with np.testing.assert_raises(ValueError):
        m.pixel_array
    pa = [r for r in g["rects"] if r[4] == 0]
    for (w, h, cx, cy, rot), (nr, nc, mean, std) in zip(pa, g["pixel_array"]):
        a = roi.RectangleROI(d["slice_i16"], width=w, height=h, center=(cx, cy)).pixel_array.cpu().numpy()
        assert a.shape == (int(nr), int(nc)) and abs(a.mean() - mean) < 1e-10 and abs(a.std() - std) < 1e-10
    fc = roi.RectangleROI.from_phantom_center(d["slice_i16"], width=14.5, height=9.25, angle=-60.0, dist_from_center=80.5,
                                              phantom_center=(250.3, 260.7), rotation=22.5)
    wfc = g["from_center"]
    assert abs(fc._xy[0] - wfc[0]) < 1e-12 and abs(fc._xy[1] - wfc[1]) < 1e-12
    assert fc._s()[0] == wfc[2] and fc.min == wfc[5] and fc.max == wfc[6] and abs(fc.mean - wfc[3]) < 1e-10
    with np.testing.assert_raises(ValueError):
        roi.RectangleROI(d["slice_i16"], width=1.5, height=5, center=(10, 10))
    # raw polygons: the pixel COUNT and the sum of an index image identify the pixel set
    shape = tuple(int(v) for v in g["poly_shape"])
    idx = np.arange(shape[0] * shape[1], dtype=np.float64).reshape(shape)
    off = g["poly_offsets"]
    for k in range(len(g["poly_nverts"])):
        nv = int(g["poly_nverts"][k])
        out, status = roi.polygon_roi_stats_batch(torch.from_numpy(idx).to(dev)[None], g["poly_vertices"][k][None, :nv])
        rr, cc = g["poly_rr"][off[k]:off[k + 1]], g["poly_cc"][off[k]:off[k + 1]]
        if len(rr) == 0:
            assert int(status[0, 0]) == 3, k
            continue
        o6 = out.cpu().numpy()[0, 0]
        vals = idx[rr, cc]
        assert int(status[0, 0]) == 0 and o6[0] == len(rr) and o6[3] == vals.min() and o6[4] == vals.max() \
            and o6[5] == np.median(vals) and abs(o6[1] - vals.mean()) < 1e-9, k


# ---------------------------------------------------------------------------------------------- Hill / penumbra
_HILL_EDGES = {"fwhm": "FWHM", "infl": "Inflection Derivative", "hill": "Inflection Hill"}


def _pen_keys(edge, lower, upper, dpmm):
    k = [f"left {lower}% index (exact)", f"left {upper}% index (exact)", f"right {lower}% index (exact)",
         f"right {upper}% index (exact)", "left penumbra width (exact)", "right penumbra width (exact)"]
    if edge == "hill":
        k += [f"left {lower}% value (exact)", f"left {upper}% value (exact)", f"right {lower}% value (exact)",
              f"right {upper}% value (exact)", "left gradient (exact)", "right gradient (exact)"]
    if edge == "fwhm":
        k += [f"left {lower}% value (@rounded)", f"right {upper}% value (@rounded)"]
    if dpmm:
        k += ["left penumbra width (exact) mm", "right penumbra width (exact) mm"]
        if edge == "hill":
            k += ["left gradient (exact) %/mm", "right gradient (exact) %/mm"]
    return k


def hill_cases(g):
    """(tag, values, edge, constructor kwargs) for every profile of tests/golden/hill.npz"""
    for i in range(int(g["n_fixtures"])):
        for edge in _HILL_EDGES:
            for mode, interp in (("none", None), ("linear", "Linear")):
                kw = dict(interpolation=interp)
                if edge == "hill":
                    kw["hill_window_ratio"] = 0.5
                else:
                    kw["x_values"] = g[f"fx{i}.x"]
                yield f"fx{i}.{edge}.{mode}", g[f"fx{i}.y"], edge, kw
    for edge in _HILL_EDGES:
        yield f"epid.{edge}.dpmm", g["epid.y"], edge, dict(dpmm=1 / 0.336)
        yield f"epid.{edge}.wide", g["epid.y"], edge, dict(dpmm=1 / 0.336, hill_window_ratio=0.2, interpolation="Spline")
    for k in range(4):
        for edge in _HILL_EDGES:
            yield f"fff{k}.{edge}.px", g[f"fff{k}.y"], edge, dict(dpmm=2.5 + k, hill_window_ratio=(0.1, 0.2, 0.3, 0.05)[k])


def check_hill_and_penumbra(g, make_profile, tol=1e-9, only=None, spline_tol=None):
    """Every profile of hill.npz through `make_profile(values, edge_name, **kw)` (the oracle class or the device
    mirror): Hill inflection data / parameters / beam centre / field data and penumbra() for the three edge methods
    against the reference's own numbers.  `tol` is relative (the fits come from the same MINPACK routine).
    `spline_tol` applies to the cubic-spline-resampled Hill cases: the device spline agrees with scipy's to ~1e-13, and
    MINPACK stops at a relative tolerance of 1.5e-8, so its stopping point -- the fitted parameters -- moves by up to
    ~1e-7 relative for such inputs (the reference's own Hill tests use deltas of 0.01-0.1)."""
    n = 0
    for tag, values, edge, kw in hill_cases(g):
        if only is not None and not only(tag):
            continue
        if f"{tag}.error" in g:
            try:
                make_profile(values.copy(), _HILL_EDGES[edge], **kw).penumbra(20, 80)
            except (ValueError, IndexError, RuntimeError):
                continue
            raise AssertionError(f"{tag}: the reference raised {g[tag + '.error']}")
        p = make_profile(values.copy(), _HILL_EDGES[edge], **kw)
        # Hill fits: MINPACK stops at a relative tolerance of 1.5e-8 and numpy's vectorised pow() is not bit-reproducible
        # from run to run (SIMD body vs scalar remainder depends on buffer alignment), so even the reference against
        # itself reproduces fitted parameters only to ~1e-8..1e-6 (ill-conditioned slope / plateau parameters): everything
        # downstream of a fit is held to 1e-5 (the reference's own Hill tests use deltas of 0.01-0.1)
        t = max(tol, spline_tol or 0.0, 1e-5) if edge == "hill" else tol

        def close(a, b, what, t=t):
            a, b = np.asarray(a, float), np.asarray(b, float)
            assert a.shape == b.shape and np.allclose(a, b, rtol=t, atol=t), (tag, what, a, b)

        close(p.values, g[f"{tag}.values"], "values", t)
        if edge == "hill":
            inf = p.inflection_data()
            close([inf[k] for k in g["hill_keys"]], g[f"{tag}.infl"], "inflection")
            assert [inf["left index (rounded)"], inf["right index (rounded)"]] == list(g[f"{tag}.infl_rounded"]), tag
            close(np.array([inf["left Hill params"], inf["right Hill params"]]), g[f"{tag}.params"], "params")
            bc = p.beam_center()
            close([bc["index (exact)"], bc["value (@rounded)"]], g[f"{tag}.beam_center"], "beam centre")
            fd = p.field_data(in_field_ratio=0.8, slope_exclusion_ratio=0.2)
            close([fd[k] for k in g["field_keys"]], g[f"{tag}.field"], "field data")
        for lower, upper in ((20, 80), (10, 90)):
            pen = p.penumbra(lower, upper)
            keys = _pen_keys(edge, lower, upper, kw.get("dpmm"))
            close([pen[k] for k in keys], g[f"{tag}.pen{lower}_{upper}"], f"penumbra {lower}/{upper}")
            close(pen["left values"], g[f"{tag}.pen{lower}_{upper}.left_values"], "left values")
            close(pen["right values"], g[f"{tag}.pen{lower}_{upper}.right_values"], "right values")
        n += 1
    return n


def check_hill_batch(g, run_batch, only=None):
    """`run_batch(values [N, L], **constructor kwargs)` -> profile.HillEdgesBatch.  The INFLECTION_HILL profiles of hill.npz,
    grouped by length and constructor arguments (a batch shares both), against the reference's own SingleProfile: the
    processed values (1e-9: identical arithmetic upstream of the fit), the inflection dictionary and both parameter sets at the
    1e-5 this row holds downstream of a fit; where the reference raised, `info` must report a failure.  A fit whose exponent
    |d| exceeds 100 is a step inside one sample spacing (curve_fit warns that the covariance cannot be estimated): its
    inflection INDEX is held to 1e-5 like the others, the curve's value there and d itself only to 1e-3 / not at all."""
    groups = {}
    for tag, values, edge, kw in hill_cases(g):
        if edge != "hill" or (only is not None and not only(tag)):
            continue
        key = (len(values), tuple(sorted((k, str(v)) for k, v in kw.items())))
        groups.setdefault(key, []).append((tag, values, kw))
    n = 0
    for items in groups.values():
        res = run_batch(np.stack([v for _, v, _ in items]), **items[0][2])
        info = res.info.cpu().numpy()
        for i, (tag, _, _) in enumerate(items):
            if f"{tag}.error" in g:
                assert not ((info[i] >= 1) & (info[i] <= 4)).all(), (tag, "the reference raised", str(g[tag + ".error"]), info[i])
                try:
                    res.inflection_data(i)
                except (ValueError, IndexError, RuntimeError, TypeError):
                    n += 1
                    continue
                raise AssertionError(f"{tag}: inflection_data() did not raise")
            assert ((info[i] >= 1) & (info[i] <= 4)).all(), (tag, info[i])
            assert np.allclose(res.values[i].cpu().numpy(), g[f"{tag}.values"], rtol=1e-9, atol=1e-9), (tag, "values")
            d = res.inflection_data(i)
            want_p = g[f"{tag}.params"]
            want = dict(zip([str(k) for k in g["hill_keys"]], g[f"{tag}.infl"]))
            for side, wp in zip(("left", "right"), want_p):
                step = abs(wp[3]) > 100
                got_p = d[f"{side} Hill params"]
                assert np.allclose(got_p[:3], wp[:3], rtol=1e-5, atol=1e-5), (tag, side, got_p, wp)
                if not step:
                    assert np.allclose(got_p[3], wp[3], rtol=1e-5, atol=1e-5), (tag, side, got_p, wp)
                assert np.isclose(d[f"{side} index (exact)"], want[f"{side} index (exact)"], rtol=1e-5, atol=1e-5), (tag, side)
                assert np.isclose(d[f"{side} value (@exact)"], want[f"{side} value (@exact)"], rtol=1e-3 if step else 1e-5,
                                  atol=1e-5), (tag, side)
            assert [d["left index (rounded)"], d["right index (rounded)"]] == list(g[f"{tag}.infl_rounded"]), tag
            n += 1
    return n


def penumbra_windows(n, seed=0, mmax=64):
    """`n` synthetic penumbra windows (rising and falling Hill curves with detector noise, 12-60 samples) -> x, y [n, mmax],
    lens [n]"""
    rng = np.random.default_rng(seed)
    xs, ys, lens = np.zeros((n, mmax)), np.zeros((n, mmax)), np.zeros(n, np.int32)
    for t in range(n):
        m = int(rng.integers(12, min(60, mmax)))
        c0 = rng.uniform(30, 400)
        x = np.arange(int(c0) - m // 2, int(c0) - m // 2 + m).astype(float)
        d = rng.uniform(8, 60) * (-1 if t % 2 == 0 else 1)
        a, b = rng.uniform(0, 0.1), rng.uniform(0.8, 1.2)
        xs[t, :m] = x
        ys[t, :m] = a + (b - a) / (1.0 + (c0 / x) ** d) + rng.normal(0, 0.004, m)
        lens[t] = m
    return xs, ys, lens


def check_hill_fit_vs_scipy(fit, n=60, seed=0):
    """`fit(x [n, M], y, lens)` -> (params, info, nfev) numpy: the device Levenberg-Marquardt against scipy's (MINPACK lmdif
    through leastsq, what curve_fit calls) on synthetic windows, through what the restatement promises: the same verdict
    (converged or not) for every window, an equally good minimum (sum of squares within 1e-4 relative), the same number of
    function evaluations for at least 90 % of the fits (the iteration follows MINPACK's path; pow()'s last bit can move a
    stopping test) and the inflection point to 1e-5 for at least 95 % of them -- the rest are windows that miss a plateau: a
    flat valley in parameter space where the stopping point (not the quality of the fit) depends on the last bit of pow();
    those are held to 2e-3.  When `fit` also returns pl_hill_fit_ex's last-step length, EVERY fit whose last accepted step
    is at most 1e-6 of the parameter vector ("settled", what HillEdgesBatch.settled reports) must meet 1e-5, and at least 90 %
    of the converged fits must be settled.  The reference's real windows are held to 1e-5 one by one in check_hill_batch."""
    import warnings

    from scipy.optimize import leastsq

    xs, ys, lens = penumbra_windows(n, seed)
    got = fit(xs, ys, lens)
    params, info, nfev = got[:3]
    step = got[3] if len(got) > 3 else None                 # pl_hill_fit_ex: the relative length of the last accepted step
    same_nfev = converged = tight = settled = settled_tight = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(n):
            x, y = xs[i, :lens[i]], ys[i, :lens[i]]
            resid = lambda p: p[0] + (p[1] - p[0]) / (1.0 + (p[2] / x) ** p[3]) - y
            ref, _, extra, _, ier = leastsq(resid, (y.min(), y.max(), np.median(x), 0), full_output=True)
            ok_ref, ok_dev = ier in (1, 2, 3, 4), 1 <= info[i] <= 4
            assert ok_ref == ok_dev, (i, ier, info[i])
            if not ok_ref:
                continue
            converged += 1
            same_nfev += int(extra["nfev"] == nfev[i])
            infl = lambda q: q[2] * ((q[3] - 1) / (q[3] + 1)) ** (1 / q[3])
            ssq_dev, ssq_ref = (resid(params[i]) ** 2).sum(), (resid(ref) ** 2).sum()
            assert abs(ssq_dev - ssq_ref) <= 1e-4 * ssq_ref, (i, ssq_dev, ssq_ref)
            assert np.isclose(infl(params[i]), infl(ref), rtol=2e-3), (i, params[i], ref)
            is_tight = bool(np.isclose(infl(params[i]), infl(ref), rtol=1e-5))
            tight += int(is_tight)
            if step is not None and step[i] <= 1.0e-6:       # HillEdgesBatch.settled: EVERY such fit is held to 1e-5
                settled += 1
                settled_tight += int(is_tight)
                assert is_tight, (i, step[i], infl(params[i]), infl(ref), extra["nfev"], nfev[i])
    assert converged >= 0.9 * n and same_nfev >= 0.9 * converged and tight >= 0.95 * converged, (converged, same_nfev, tight)
    if step is not None:                                      # the flag is not an excuse: at least nine fits in ten are settled
        assert settled >= 0.9 * converged and settled_tight == settled, (converged, settled, settled_tight)
    return converged


def check_hill_fit_kernels_agree(fit, n=40, seed=5):
    """pl_hill_fit has two kernels: eight lanes per fit with the vectors in LDS (windows of up to 126 samples) and one lane per
    fit on a global workspace (longer ones).  The same windows through both -- the second time padded to a 160-sample
    capacity -- must give IDENTICAL parameters, info and function-evaluation counts: the group kernel's leader runs the same
    operations in the same order and the model values do not depend on the lane that computes them."""
    xs, ys, lens = penumbra_windows(n, seed)
    lens[:3] = (3, 4, 5)                                     # fewer samples than parameters (info -1), and the smallest fits
    pa, ia, na = fit(xs, ys, lens)
    wide = lambda a: np.concatenate([a, np.zeros((n, 160 - a.shape[1]))], axis=1)
    pb, ib, nb = fit(wide(xs), wide(ys), lens)
    assert ia[0] == -1 and np.isnan(pa[0]).all()
    assert np.array_equal(ia, ib) and np.array_equal(na, nb), (ia, ib, na, nb)
    assert np.array_equal(pa, pb, equal_nan=True)
    return int(((ia >= 1) & (ia <= 4)).sum())


def check_hill_fit_pathological(fit, fit_ex=None):
    """Windows no fit should be asked about -- constant, NaN, infinity, 1e300, 1e-300, pure noise, a step, a zero abscissa
    (c / x = inf): both kernels of pl_hill_fit must TERMINATE, agree bit for bit, report NaN / inf as info -4 (curve_fit's
    check_finite raises ValueError there) and leave the well-posed row alone."""
    m = 24
    x = np.arange(100, 100 + m, dtype=float)
    good = 1 / (1 + (110 / x) ** 20)
    rows = [np.full(m, 0.5), np.where(np.arange(m) == 5, np.nan, good), np.where(np.arange(m) == 7, np.inf, good), 1e300 * good,
            1e-300 * good, np.random.default_rng(0).random(m), (x > 111).astype(float), good]
    ys = np.stack(rows)
    lens = np.full(len(rows), m, np.int32)
    for zero_x in (False, True):
        xs = np.stack([x] * len(rows))
        if zero_x:
            xs[:, 0] = 0.0
        outs = []
        for pad in (0, 140):                                 # the group kernel / the one-lane kernel
            grow = lambda a: np.concatenate([a, np.zeros((len(a), pad))], axis=1)
            outs.append(fit(grow(xs), grow(ys), lens))
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b, equal_nan=True)
        params, info, nfev = outs[0]
        assert info[1] == -4 and info[2] == -4 and np.isnan(params[1]).all() and np.isnan(params[2]).all(), info
        assert (nfev <= 1005).all()
        if not zero_x:
            assert 1 <= info[-1] <= 4 and abs(params[-1][2] - 110) < 1e-6 and abs(params[-1][3] - 20) < 1e-5, (info, params[-1])
        if fit_ex is not None:
            # the last accepted step of a degenerate window (all scaled parameters zero) is 0 or +inf, never NaN;
            # NaN marks only the windows that were refused (info -4), and the well-posed row is settled
            for pad in (0, 140):
                grow = lambda a: np.concatenate([a, np.zeros((len(a), pad))], axis=1)
                _, info_e, _, step = fit_ex(grow(xs), grow(ys), lens)
                ok = info_e != -4
                assert not np.isnan(step[ok]).any() and (step[ok] >= 0).all(), (step, info_e)
                if not zero_x:
                    assert step[-1] < 1e-6, step
    return True


def beam_profiles(n, length=200, seed=0):
    """`n` synthetic open-field profiles of `length` detectors: two Hill penumbrae, a slightly domed top, detector noise"""
    rng = np.random.default_rng(seed)
    x = np.arange(length, dtype=float) + 1.0
    out = np.empty((n, length))
    for i in range(n):
        left, right = rng.uniform(0.2, 0.3) * length, rng.uniform(0.7, 0.8) * length
        steep = rng.uniform(15, 40)
        rise = 1.0 / (1.0 + (left / x) ** steep)
        fall = 1.0 / (1.0 + (x / right) ** (steep * right / left))
        dome = 1.0 - rng.uniform(0, 0.05) * ((x - (left + right) / 2) / length) ** 2
        out[i] = rng.uniform(50, 200) * rise * fall * dome + rng.uniform(0, 2) + rng.normal(0, 0.05, length)
    return out


def check_hill_batch_vs_single(run_batch, make_single, n=6, length=120, sample=None, **kw):
    """The batched INFLECTION_HILL path against the per-profile SingleProfile mirror (scipy's curve_fit on the host) on the
    same synthetic profiles: processed values bit for bit (same kernels), inflection data at 1e-5"""
    profs = beam_profiles(n, length)
    res = run_batch(profs, **kw)
    info = res.info.cpu().numpy()
    vals = res.values.cpu().numpy()
    for i in (range(n) if sample is None else sample):
        single = make_single(profs[i].copy(), **kw)
        assert ((info[i] >= 1) & (info[i] <= 4)).all(), (i, info[i])
        assert np.allclose(vals[i], single.values, rtol=1e-9, atol=1e-12), i
        want, got = single.inflection_data(), res.inflection_data(i)
        for k in ("left index (exact)", "right index (exact)", "left value (@exact)", "right value (@exact)"):
            assert np.isclose(got[k], want[k], rtol=1e-5, atol=1e-5), (i, k, got[k], want[k])
    return res


def check_hill_batch_options(run_batch, make_single, length=90):
    """single_profile_hill_batch against the per-profile mirror for every normalisation / interpolation choice, with a row that
    has no field at all (a ramp: the derivative never reaches 0.8 of its range on both sides -> the reference raises, the batch
    reports it in `info` and keeps the other rows)."""
    profs = beam_profiles(4, length, seed=11)
    profs[2] = np.linspace(1.0, 2.0, length) ** 2           # monotone: no falling edge
    n_ok = 0
    for norm in (None, "Max", "Geometric center", "Beam center"):
        for interp, kw in (("Linear", {}), (None, {}), ("Spline", dict(dpmm=2.0, interpolation_resolution_mm=0.1))):
            opts = dict(normalization_method=norm, interpolation=interp, hill_window_ratio=0.3, **kw)
            res = run_batch(profs, **opts)
            info = res.info.cpu().numpy()
            for i in range(len(profs)):
                try:
                    single = make_single(profs[i].copy(), **opts)
                    want = single.inflection_data()
                except (IndexError, ValueError, RuntimeError, TypeError):
                    assert not ((info[i] >= 1) & (info[i] <= 4)).all(), (norm, interp, i, info[i])
                    continue
                assert ((info[i] >= 1) & (info[i] <= 4)).all(), (norm, interp, i, info[i])
                tol = 1e-6 if interp == "Spline" else 1e-9           # the device spline agrees with scipy's to ~1e-13
                assert np.allclose(res.values[i].cpu().numpy(), single.values, rtol=tol, atol=1e-12), (norm, interp, i)
                got = res.inflection_data(i)
                for k in ("left index (exact)", "right index (exact)", "left value (@exact)", "right value (@exact)"):
                    assert np.isclose(got[k], want[k], rtol=1e-5, atol=1e-5), (norm, interp, i, k, got[k], want[k])
                for lower, upper in ((20, 80), (10, 90)):        # penumbra(): every key but the ragged value slices
                    pen_want, pen_got = single.penumbra(lower, upper), res.penumbra(lower, upper)
                    keys = [k for k in pen_want if not k.endswith("values")]
                    assert sorted(keys) == sorted(pen_got), (sorted(keys), sorted(pen_got))
                    for k in keys:
                        assert np.isclose(float(pen_got[k][i]), pen_want[k], rtol=1e-5, atol=1e-5), (norm, interp, i, k)
                n_ok += 1
    return n_ok


def check_fwhm_batch(run_batch, make_single, length=90, xs=(50, 20, 80), norms=(None, "Max", "Geometric center", "Beam center")):
    """single_profile_fwhm_batch against the per-profile mirror (default FWHM edge method) for every normalisation /
    interpolation choice: processed values and every scalar key of fwxm_data(x) for x = 50, 20, 80; a row without a peak (a
    ramp) holds NaN where the reference raises IndexError."""
    profs = beam_profiles(4, length, seed=13)
    profs[1] = np.linspace(1.0, 2.0, length)                # monotone: find_peaks finds nothing
    n_ok = 0
    for norm in norms:
        for interp, kw in (("Linear", {}), (None, {}), ("Spline", dict(dpmm=2.0, interpolation_resolution_mm=0.1))):
            opts = dict(normalization_method=norm, interpolation=interp, **kw)
            res = run_batch(profs, **opts)
            for x in xs:
                got = {k: v.cpu().numpy() for k, v in res.fwxm_data(x).items()}
                for i in range(len(profs)):
                    try:
                        single = make_single(profs[i].copy(), **opts)
                        want = single.fwxm_data(x)
                    except IndexError:
                        assert got["peaks"][i] == 0 or np.isnan(res.values[i].cpu().numpy()).any(), (norm, interp, x, i)
                        continue
                    tol = 1e-6 if interp == "Spline" else 1e-9
                    assert np.allclose(res.values[i].cpu().numpy(), single.values, rtol=tol, atol=1e-12), (norm, interp, i)
                    keys = [k for k in want if k not in ("field values", "peak_props")]
                    assert sorted(keys) == sorted(k for k in got if k != "peaks"), (sorted(keys), sorted(got))
                    for k in keys:
                        assert np.isclose(got[k][i], want[k], rtol=tol, atol=1e-9), (norm, interp, x, i, k, got[k][i], want[k])
                    n_ok += 1
            pen = {k: v.cpu().numpy() for k, v in res.penumbra(20, 80).items()}
            for i in (0, 2, 3):                             # (row 1 has no peak)
                want = make_single(profs[i].copy(), **opts).penumbra(20, 80)
                keys = [k for k in want if not k.endswith("values")]
                assert sorted(keys) == sorted(pen), (sorted(keys), sorted(pen))
                for k in keys:
                    assert np.isclose(pen[k][i], want[k], rtol=1e-6 if interp == "Spline" else 1e-9, atol=1e-9), (norm, interp, i, k)
    return n_ok


def check_inflection_batch(run_batch, make_single, length=90, norms=(None, "Max", "Geometric center", "Beam center")):
    """single_profile_inflection_batch against the per-profile mirror (INFLECTION_DERIVATIVE) for every normalisation /
    interpolation choice, a row without a falling edge included (NaN + status 2 where the reference raises)."""
    profs = beam_profiles(4, length, seed=17)
    profs[3] = np.linspace(1.0, 2.0, length) ** 2
    n_ok = 0
    for norm in norms:
        for interp, kw in (("Linear", {}), (None, {}), ("Spline", dict(dpmm=2.0, interpolation_resolution_mm=0.1))):
            opts = dict(normalization_method=norm, interpolation=interp, **kw)
            res = run_batch(profs, **opts)
            got = {k: v.cpu().numpy() for k, v in res.inflection_data().items()}
            status = res.status.cpu().numpy()
            for i in range(len(profs)):
                try:
                    single = make_single(profs[i].copy(), **opts)
                    want = single.inflection_data()
                except (IndexError, ValueError):
                    assert status[i] != 0 and np.isnan(got["left index (exact)"][i]), (norm, interp, i)
                    continue
                tol = 1e-6 if interp == "Spline" else 1e-9
                assert status[i] == 0
                assert np.allclose(res.values[i].cpu().numpy(), single.values, rtol=tol, atol=1e-12), (norm, interp, i)
                assert sorted(want) == sorted(got)
                for k in want:
                    assert np.isclose(got[k][i], want[k], rtol=tol, atol=1e-9), (norm, interp, i, k, got[k][i], want[k])
                n_ok += 1
    return n_ok


def check_fwhm_batch_golden(g, run_batch, only_lengths=None):
    """single_profile_fwhm_batch against the REFERENCE's own SingleProfile numbers (tests/golden/single_profile.npz: its 20 frozen
    detector profiles without abscissae x 3 resampling modes, grouped by length -- a batch shares length and arguments -- and
    the EPID profile with dpmm / factor / normalisation options): x_indices and processed values (bit-identical for NONE /
    LINEAR, 1e-10 for SPLINE) and the seven scalars of fwxm_data(50 / 25 / 80) at 1e-9."""
    wkeys = [str(k) for k in g["fwxm_keys"]]
    n = 0
    groups = {}
    for i in range(20):
        groups.setdefault(len(g[f"fx{i}.y"]), []).append(i)
    cases = []
    for length, idxs in groups.items():
        if only_lengths is not None and length not in only_lengths:
            continue
        for mode, interp in (("none_nox", None), ("linear_nox", "Linear"), ("spline_nox", "Spline")):
            cases.append(([f"fx{i}.{mode}" for i in idxs], np.stack([g[f"fx{i}.y"] for i in idxs]), dict(interpolation=interp)))
    epid = {"dpmm": dict(dpmm=1 / 0.336), "dpmm_spline": dict(dpmm=1 / 0.336, interpolation="Spline", interpolation_resolution_mm=0.05),
            "factor3": dict(interpolation_factor=3), "max": dict(normalization_method="Max"),
            "geo": dict(normalization_method="Geometric center"), "raw": dict(normalization_method=None, ground=False, interpolation=None)}
    if only_lengths is None:
        for name, kw in epid.items():
            cases.append(([f"epid.{name}"], g["epid.y"][None].copy(), kw))
    for tags, values, kw in cases:
        res = run_batch(values, **kw)
        vtol = 1e-10 if kw.get("interpolation") == "Spline" else 0
        got = {h: {k: v.cpu().numpy() for k, v in res.fwxm_data(h).items()} for h in (50, 25, 80)}
        vals = res.values.cpu().numpy()
        for r, tag in enumerate(tags):
            assert np.array_equal(np.asarray(res.x_indices, float), g[f"{tag}.x_indices"]), tag
            assert np.allclose(vals[r], g[f"{tag}.values"], rtol=vtol, atol=vtol), tag
            for h in (50, 25, 80):
                assert np.allclose([got[h][k][r] for k in wkeys], g[f"{tag}.fwxm{h}"], rtol=1e-9, atol=1e-9), (tag, h)
            n += 1
    return n


def check_profile_batch_golden(g, batch_fns):
    """The three batched SingleProfile paths against the REFERENCE's own SingleProfile (tests/golden/profile_batch.npz, made by
    tests/golden/make_profile_batch_golden.py: its frozen 63-detector profiles and synthetic open-field profiles, three
    constructor option sets): `batch_fns` = dict(infl=, fwhm=, hill=) -> the batch objects.  inflection_data() and the
    processed values of the derivative method, penumbra(20, 80) of the FWHM and Hill methods; 1e-9 (1e-5 downstream of a Hill
    fit); where the reference raised, the batch must flag the row."""
    opts = {"default": {}, "none": dict(interpolation=None), "max_dpmm": dict(normalization_method="Max", dpmm=2.0)}
    ikeys, fkeys, hkeys = ([str(k) for k in g[n]] for n in ("infl_keys", "fwhm_pen_keys", "hill_pen_keys"))
    n = 0
    for sname in g["sets"]:
        rows = g[f"{sname}.rows"]
        for oname, kw in opts.items():
            infl = batch_fns["infl"](rows, **kw)
            idata = {k: v.cpu().numpy() for k, v in infl.inflection_data().items()}
            ivals, istat = infl.values.cpu().numpy(), infl.status.cpu().numpy()
            fpen = {k: v.cpu().numpy() for k, v in batch_fns["fwhm"](rows, **kw).penumbra(20, 80).items()}
            hill = batch_fns["hill"](rows, **kw)
            hpen = {k: v.cpu().numpy() for k, v in hill.penumbra(20, 80).items()}
            hinfo = hill.info.cpu().numpy()
            for r in range(len(rows)):
                tag = f"{sname}.{oname}.{r}"
                if f"{tag}.infl.error" in g:
                    assert istat[r] != 0, tag
                else:
                    assert istat[r] == 0, tag
                    assert np.allclose(ivals[r], g[f"{tag}.infl_values"], rtol=1e-9, atol=1e-12), tag
                    assert np.allclose([idata[k][r] for k in ikeys], g[f"{tag}.infl"], rtol=1e-9, atol=1e-9), tag
                if f"{tag}.fwhm.error" not in g:
                    assert np.allclose([fpen[k][r] for k in fkeys], g[f"{tag}.fwhm_pen"], rtol=1e-9, atol=1e-9), tag
                if f"{tag}.hill.error" in g:
                    assert not ((hinfo[r] >= 1) & (hinfo[r] <= 4)).all() or np.isnan([hpen[k][r] for k in hkeys]).any(), tag
                else:
                    assert ((hinfo[r] >= 1) & (hinfo[r] <= 4)).all(), (tag, hinfo[r])
                    got, want = np.array([hpen[k][r] for k in hkeys]), g[f"{tag}.hill_pen"]
                    assert np.allclose(got, want, rtol=1e-5, atol=1e-5), (tag, got, want)
                n += 1
    return n


# ---------------------------------------------------------------------------------------------- Starshot
def starshot_cases(g):
    for name in g["names"]:
        name = str(name)
        frame = g["four.frame"] if name == "startpt" else g[f"{name}.frame"]
        yield name, frame, float(g[f"{name}.dpi"]), eval(str(g[f"{name}.kw"]), {"__builtins__": {}}, {})


def check_starshot(g, make, only=None, tol=1e-9):
    """`make(frame, dpi, sid)` -> an object with the reference's Starshot interface; compared with the reference's own
    Starshot.analyze() on the synthetic frames of tests/golden/starshot.npz: automatic start point and local maximum,
    the rolled / filtered / grounded star profile, peak indices / values / image coordinates, the paired lines, wobble
    centre / radius (the Nelder-Mead fit sees identical lines, so identical iterates), angles, pass flag."""
    n = 0
    for name, frame, dpi, kw in starshot_cases(g):
        if only and name not in only:
            continue
        s = make(frame.copy(), dpi, 1000)
        s.analyze(**kw)
        sp, local_max = s._get_reasonable_start_point()
        want = g[f"{name}.start"]
        assert (sp.x, sp.y) == (want[0], want[1]) and abs(local_max - want[2]) <= 1e-6 * abs(want[2]), (name, "start")
        cp = s.circle_profile
        assert np.allclose([cp.center.x, cp.center.y, cp.radius], g[f"{name}.circle"], rtol=0, atol=1e-12), name
        assert np.allclose(np.asarray(cp.values, float), g[f"{name}.profile"], rtol=tol, atol=tol), (name, "profile")
        peaks = np.array([[p.idx, p.value, p.x, p.y] for p in cp.peaks], dtype=float)
        assert peaks.shape == g[f"{name}.peaks"].shape and np.allclose(peaks, g[f"{name}.peaks"], rtol=tol, atol=tol), name
        lines = np.array([[ln.point1.x, ln.point1.y, ln.point2.x, ln.point2.y] for ln in s.lines.lines], dtype=float)
        assert np.allclose(lines, g[f"{name}.lines"], rtol=tol, atol=tol), (name, "lines")
        w = [s.wobble.center.x, s.wobble.center.y, s.wobble.radius, s.wobble.radius_mm, s.wobble.diameter_mm]
        # Nelder-Mead (fatol 1e-3, xatol 1e-4) retraces the reference's simplex exactly when the lines are bit-identical;
        # lines that differ in the last bits may stop a few 1e-4 px away -- its own termination tolerance
        wtol = 1e-7 if np.array_equal(lines, g[f"{name}.lines"]) else 2e-3
        assert np.allclose(w, g[f"{name}.wobble"], rtol=wtol, atol=wtol), (name, "wobble", w, g[f"{name}.wobble"])
        assert np.allclose(s.angles, g[f"{name}.angles"], rtol=1e-9, atol=1e-9), (name, "angles")
        assert s.passed == bool(g[f"{name}.passed"]), name
        n += 1
    return n


def check_starshot_batch(g, dev, names=("four", "six_off", "inverted", "nofwhm"), variants=3):
    """starshot.analyze_batch against (a) the reference's own Starshot.analyze() numbers for the golden frame that leads each
    stack and (b) the class API, frame by frame, for shifted / histogram-inverted copies of it (other start points, so other
    ring sizes and more than one gather group per pass)."""
    from pylinac_amd import starshot

    n_checked = 0
    for name, frame, dpi, kw in starshot_cases(g):
        if name not in names or frame.dtype != np.uint16:
            continue
        stack = [frame]
        for k in range(1, variants):
            v = np.roll(frame, (2 * k + 1, -3 * k), axis=(0, 1))
            if k % 2 == 0:
                v = (int(v.max()) + int(v.min()) - v.astype(np.int64)).astype(np.uint16)      # an inverted film
            stack.append(v)
        stack = np.stack(stack)
        kw = dict(kw)
        kw.pop("start_point", None)
        res = starshot.analyze_batch(torch.from_numpy(stack).to(dev), dpi=dpi, sid=1000, **kw)
        assert len(res) == len(stack) and (res.status == 0).all(), (name, res.status)
        want = g[f"{name}.wobble"]
        got = [res.wobble_center[0, 0], res.wobble_center[0, 1], res.wobble_radius[0], res.wobble_radius_mm[0],
               res.wobble_diameter_mm[0]]
        a0 = res.analyzers[0]
        lines = np.array([[ln.point1.x, ln.point1.y, ln.point2.x, ln.point2.y] for ln in a0.lines.lines], dtype=float)
        assert np.allclose(lines, g[f"{name}.lines"], rtol=1e-9, atol=1e-9), (name, "lines")
        wtol = 1e-7 if np.array_equal(lines, g[f"{name}.lines"]) else 2e-3
        assert np.allclose(got, want, rtol=wtol, atol=wtol), (name, got, want)
        assert np.allclose(np.asarray(a0.circle_profile.values, float), g[f"{name}.profile"], rtol=1e-9, atol=1e-9), name
        assert tuple(res.start_point[0]) == tuple(g[f"{name}.start"][:2]) and bool(res.passed[0]) == bool(g[f"{name}.passed"])
        assert np.allclose(a0.angles, g[f"{name}.angles"], rtol=1e-9, atol=1e-9), name
        for i in range(len(stack)):
            s1 = starshot.Starshot(stack[i].copy(), dpi=dpi, sid=1000)
            s1.analyze(**kw)
            sp, lm = s1._get_reasonable_start_point()
            assert (sp.x, sp.y) == tuple(res.start_point[i]) and lm == res.local_max[i], (name, i, "start")
            assert np.array_equal(np.asarray(s1.circle_profile.values), np.asarray(res.analyzers[i].circle_profile.values)), (name, i)
            assert [(q.idx, q.x, q.y) for q in s1.circle_profile.peaks] == \
                   [(q.idx, q.x, q.y) for q in res.analyzers[i].circle_profile.peaks], (name, i, "peaks")
            assert (s1.wobble.center.x, s1.wobble.center.y, s1.wobble.radius) == \
                   (res.wobble_center[i, 0], res.wobble_center[i, 1], res.wobble_radius[i]), (name, i, "wobble")
            assert s1.passed == bool(res.passed[i]) and len(s1.lines) == res.n_lines[i]
            n_checked += 1
    # the error cases come back as codes: a flat frame has no FW80M peak (IndexError in the class), a frame without spokes
    # exhausts the sweep (RuntimeError in the class), and with recursive=False it fails at the first peak count
    frame = g["four.frame"]
    flat = np.full_like(frame, 1000)
    rng = np.random.default_rng(5)
    blob = (1000 + 20000 * np.exp(-(((np.arange(frame.shape[0])[:, None] - 300) ** 2 + (np.arange(frame.shape[1])[None, :] - 320) ** 2)
                                    / (2 * 60.0 ** 2)))).astype(np.uint16)
    blob += rng.integers(0, 3, blob.shape).astype(np.uint16)
    res = starshot.analyze_batch(torch.from_numpy(np.stack([frame, flat, blob])).to(dev), dpi=100, sid=1000)
    assert res.status.tolist() == [0, 3, 1], res.status
    assert np.isnan(res.wobble_radius[1:]).all() and res.analyzers[1] is None and res.analyzers[2] is None
    res = starshot.analyze_batch(torch.from_numpy(np.stack([frame, blob])).to(dev), dpi=100, sid=1000, recursive=False)
    assert res.status.tolist() == [0, 2], res.status
    return n_checked


# ---------------------------------------------------------------------------------------------- contrast ROIs
def check_contrast_rois(golden, dev):
    """LowContrastDiskROI / HighContrastDiskROI and the pylinac.core.contrast formulas against the reference's own
    classes (tests/golden/contrast.npz).  Median, min, max exact; std / derived ratios to 1e-12."""
    from pylinac_amd import contrast as con
    from pylinac_amd import roi

    g, d = golden("contrast"), golden("roi")
    arr = d["slice_i16"].astype(np.float64) + float(g["shift"])
    frame = torch.from_numpy(arr).to(dev)
    props = [str(p) for p in g["props"]]
    for ci, (cx, cy, r, ref) in enumerate(g["cases"]):
        for mi, m in enumerate(g["methods"]):
            z = roi.LowContrastDiskROI(frame, radius=r, center=(cx, cy), contrast_threshold=0.01, contrast_reference=ref,
                                       cnr_threshold=0.5, contrast_method=str(m), visibility_threshold=0.1)
            got = np.array([float(getattr(z, p)) for p in props])
            np.testing.assert_allclose(got, g["low"][ci, mi], rtol=1e-12, atol=0, err_msg=f"{ci} {m}")
    fc = roi.LowContrastDiskROI.from_phantom_center(frame, angle=-33.0, roi_radius=6.5, dist_from_center=70.25,
                                                    phantom_center=(250.3, 260.7), contrast_threshold=0.02,
                                                    contrast_reference=1190.0, cnr_threshold=1.0)
    got = [fc._xy[0], fc._xy[1], fc.pixel_value, fc.contrast, fc.visibility, float(fc.passed)]
    np.testing.assert_allclose(got, g["low_from_center"], rtol=1e-12)
    hc = roi.HighContrastDiskROI(frame, radius=9.0, center=(200.5, 310.25), contrast_threshold=0.5)
    np.testing.assert_allclose([hc.max, hc.min, hc.pixel_value, hc.std], g["high"], rtol=1e-12)
    v = g["fn_in"]
    fn = [con.michelson(v), con.rms(v), con.weber(3.0, 2.0), con.ratio(3.0, 2.0), con.difference(3.0, 5.5),
          con.contrast(np.array([3.0, 2.0]), "Weber"), con.contrast(v, con.Contrast.RMS),
          con.visibility(np.array([3.0, 2.0]), 5.0, 0.7, "Michelson")]
    assert np.array_equal(fn, g["fn"])
    for bad in ("Weber", "Ratio", "Difference"):
        with np.testing.assert_raises(ValueError):
            con.contrast(v, bad)
    with np.testing.assert_raises(ValueError):
        con.contrast(v, "nope")
    with np.testing.assert_raises(ValueError):
        con.rms(np.array([0.5, 1.5]))


def check_canny_integer_images(golden, dev):
    """canny on uint8 / uint16 / int16 images (img_as_float scaling, dtype_max-scaled absolute thresholds) against
    scikit-image 0.18.3's own feature.canny: identical edge maps."""
    from pylinac_amd import canny

    g = golden("canny_int")
    for n in g["names"]:
        kw = eval(str(g[f"{n}.kw"]), {"__builtins__": {}}, {"dict": dict})
        got = canny.canny(torch.from_numpy(g[f"{n}.img"]).to(dev), **kw)
        assert np.array_equal(got.cpu().numpy().astype(bool), g[f"{n}.edges"]), n


def check_canny_masked(golden, dev):
    """canny(mask=...) against scikit-image 0.18.3's own feature.canny (tests/golden/skimage_canny_mask_py39.py): disk, half-plane,
    speckle and border masks, float64 and uint16 images, an all-false mask; plus the batched form with one mask per frame."""
    from pylinac_amd import canny

    g = golden("canny_mask")
    for n in g["names"]:
        kw = eval(str(g[f"{n}.kw"]), {"__builtins__": {}}, {"dict": dict})
        got = canny.canny(torch.from_numpy(g[f"{n}.img"]).to(dev), mask=g[f"{n}.mask"], **kw)
        assert np.array_equal(got.cpu().numpy().astype(bool), g[f"{n}.edges"]), n
    # two frames of one shape, each with its own mask, in ONE call == the two single calls
    a, b = "f64_half", "f64_half"
    img = np.stack([g["f64_half.img"], g["f64_half.img"][::-1].copy()])
    msk = np.stack([g["f64_half.mask"], g["f64_half.mask"][:, ::-1].copy()])
    kw = eval(str(g["f64_half.kw"]), {"__builtins__": {}}, {"dict": dict})
    both = canny.canny(torch.from_numpy(img).to(dev), mask=msk, **kw).cpu().numpy()
    for k in range(2):
        one = canny.canny(torch.from_numpy(img[k]).to(dev), mask=msk[k], **kw).cpu().numpy()
        assert np.array_equal(both[k], one), k
    assert np.array_equal(both[0].astype(bool), g["f64_half.edges"])


def check_rescale_dicom_values(dev):
    """image.rescale_dicom_values against the oracle restatement (pydicom's apply_rescale cannot be run here) and, as the
    pin, against the known answers of the reference's own tests for it (tests_basic/core/test_image.py:131-229): uint16 / int16
    stored values, CT-like and EPID-like tags, forced / tag-driven / suppressed inversion, no overflow when inverting."""
    from oracle import pylinac_oracle as o
    from pylinac_amd import image

    rng = np.random.default_rng(31)
    for arr in (rng.integers(0, 65535, (2, 33, 47)).astype(np.uint16), rng.integers(-2000, 3000, (2, 20, 64)).astype(np.int16)):
        for kw in (dict(rescale_slope=1.0, rescale_intercept=-1024.0), dict(rescale_slope=0.00036621, rescale_intercept=-3.7e-5,
                                                                            pixel_intensity_relationship_sign=-1),
                   dict(rescale_slope=2.5, rescale_intercept=10.0, pixel_intensity_relationship_sign=-1, invert_pixels=False),
                   dict(invert_pixels=True), dict(), dict(rescale_slope=3.0, rescale_intercept=1.0, raw_pixels=True)):
            got = image.rescale_dicom_values(torch.from_numpy(arr).to(dev), **kw).cpu().numpy()
            want = np.stack([o.rescale_dicom_values(f, **kw) for f in arr])
            assert got.dtype == want.dtype and np.array_equal(got, want), (arr.dtype, kw)
    # ---- the identities the reference's OWN tests state (tests_basic/core/test_image.py:131-200), as known answers:
    # raw_pixels=True and "no tags" leave the array alone (:131-147); with both tags the result is
    # RescaleSlope * pixel_array + RescaleIntercept (:160-170); with PixelIntensityRelationshipSign = +1 the automatic
    # choice equals the forced NON-inversion and differs from the forced inversion, with -1 it equals the forced
    # inversion (:172-200); the inversion is `max - a + min` (image.py:384-388).
    a = rng.integers(0, 4096, (1, 9, 11)).astype(np.uint16)
    t = torch.from_numpy(a).to(dev)
    same = lambda x, y: np.array_equal(x.cpu().numpy(), y)
    assert same(image.rescale_dicom_values(t, rescale_slope=2.0, rescale_intercept=5.0, raw_pixels=True), a)
    assert same(image.rescale_dicom_values(t), a)
    assert same(image.rescale_dicom_values(t, rescale_slope=1.5, rescale_intercept=-1000.0), 1.5 * a + -1000.0)
    ones = np.ones((1, 3, 3)); ones[0, 0, 0] = 100                       # the reference's own 3x3 case
    to = torch.from_numpy(ones).to(dev)
    for sign, auto_equals_forced_inversion in ((1, False), (-1, True)):
        kw = dict(rescale_slope=1, rescale_intercept=-1000, pixel_intensity_relationship_sign=sign)
        forced = image.rescale_dicom_values(to, invert_pixels=True, **kw).cpu().numpy()
        plain = image.rescale_dicom_values(to, invert_pixels=False, **kw).cpu().numpy()
        auto = image.rescale_dicom_values(to, invert_pixels=None, **kw).cpu().numpy()
        scaled = 1 * ones + -1000
        assert np.array_equal(plain, scaled) and np.array_equal(forced, scaled.max() - scaled + scaled.min())
        assert np.array_equal(auto, forced if auto_equals_forced_inversion else plain)
        assert not np.array_equal(forced, plain)
    # test_no_overflow_when_inverting (tests_basic/core/test_image.py:210-229): arrays whose min + max exceed their dtype's range
    # come back swapped, in their own dtype (the reference's parameters; its int8 case is a dtype no DICOM pixel format and no
    # entry point of this library has: refused loudly)
    for arr in (np.array([200, 250], dtype=np.uint8), np.array([60_000, 60_000], dtype=np.uint16),
                np.array([2**31 - 100, 2**31 - 1], dtype=np.int32)):
        inv = image.rescale_dicom_values(torch.from_numpy(arr.reshape(1, 1, 2)).to(dev), invert_pixels=True, raw_pixels=False)
        inv = inv.cpu().numpy().ravel()
        assert inv.dtype == arr.dtype and inv[0] == arr[1] and inv[1] == arr[0], (arr.dtype, inv)
    try:
        image.rescale_dicom_values(torch.from_numpy(np.array([120, 127], dtype=np.int8).reshape(1, 1, 2)).to(dev), invert_pixels=True)
        raise AssertionError("int8 frames must be refused")
    except TypeError:
        pass


def check_thickness_roi(golden, dev):
    """ThicknessROI.wire_fwhm / long_profile against the reference's own pylinac.ct.ThicknessROI (int16 and float64 slices,
    wires along x and along y)."""
    from pylinac_amd import roi

    g = golden("thickness")
    for k, (cx, cy, width, height) in enumerate(g["specs"]):
        r = roi.ThicknessROI(torch.from_numpy(g[f"img{k}"]).to(dev), width=width, height=height, center=(cx, cy))
        prof = r.long_profile
        if g[f"img{k}"].dtype.kind == "i":      # integer windows: the truncating Gaussian is bit-exact
            assert np.array_equal(np.asarray(prof.values, float), g[f"profile{k}"]), k
        else:                                   # float windows: same taps, summation order may differ in the last ulp
            assert np.allclose(np.asarray(prof.values, float), g[f"profile{k}"], rtol=1e-13, atol=0), k
        want = g["results"][k]
        assert abs(r.wire_fwhm - want[0]) <= 1e-9 * want[0] and len(prof.values) == int(want[1]), (k, r.wire_fwhm, want)


def check_field_strips(golden, dev):
    """Strip profiles and the centre search of FieldAnalysis against the reference's own methods
    (tests/golden/field_strips.npz): edges identical, uint16 profiles identical, float64 profiles to 1e-12, centre
    ratios to 1e-9."""
    from pylinac_amd import field_analysis as fa

    g = golden("field_strips")
    for name in ("frames", "frames_f64"):
        frames = torch.from_numpy(g[name]).to(dev)
        for k, (pos, width) in enumerate(g["specs"]):
            hv, b, t = fa.horiz_values(frames, pos, width)
            vv, left, right = fa.vert_values(frames, pos, width)
            for i in range(frames.shape[0]):
                assert [b, t, left, right] == list(g[f"{name}.{i}.edges{k}"]), (name, k)
                for got, key in ((hv[i], "h"), (vv[i], "v")):
                    want = g[f"{name}.{i}.{key}{k}"]
                    got = got.cpu().numpy()
                    if name == "frames":
                        assert np.array_equal(got, want), (name, i, key, k)
                    else:
                        assert np.allclose(got, want, rtol=1e-12, atol=0), (name, i, key, k)
        for i in range(frames.shape[0]):
            for cname, c in (("beam", "Beam center"), ("geo", "Geometric center")):
                got = fa.determine_center(frames[i], c)
                assert np.allclose(got, g[f"{name}.{i}.center_{cname}"], rtol=1e-9, atol=1e-12), (name, i, cname, got)


# ---------------------------------------------------------------------------------------------- new-style edge profiles
def edge_profile_cases(g):
    """(tag, kind, values, kwargs) for tests/golden/edge_profiles.npz; kwargs in the reference's constructor names"""
    for i in range(20):
        x, y = g[f"fx{i}.x"], g[f"fx{i}.y"]
        yield f"fx{i}.infl", "infl", y, dict(x_values=x)
        yield f"fx{i}.infl_ground_max", "infl", y, dict(x_values=x, ground=True, normalization="Max", edge_smoothing_ratio=0.01)
        yield f"fx{i}.hill", "hill", y, dict(x_values=x, hill_window_ratio=0.3)
    yield "epid.infl", "infl", g["epid.y"], dict()
    yield "epid.hill", "hill", g["epid.y"], dict()
    yield "epid.hill_beam", "hill", g["epid.y"], dict(normalization="Beam center", hill_window_ratio=0.2)
    for k in range(3):
        yield f"fff{k}.infl", "infl", g[f"fff{k}.y"], dict(x_values=g[f"fff{k}.x"])
        yield f"fff{k}.hill", "hill", g[f"fff{k}.y"], dict(x_values=g[f"fff{k}.x"], hill_window_ratio=0.15)


def check_edge_profile_known_answers(make):
    """The reference's literal known answers for the inflection / Hill profile classes
    (tests_basic/core/test_profile.py:383-448; deltas 0.01 / 0.1 as there)."""
    s21 = np.array([0, 1, 2, 4, 6, 8, 9, 10, 10, 10, 10, 10, 10, 10, 9, 8, 6, 4, 2, 1, 0], dtype=float)
    s20 = np.array([0, 1, 2, 4, 6, 8, 9, 10, 10, 10, 10, 10, 10, 9, 8, 6, 4, 2, 1, 0], dtype=float)
    sharp21 = np.array([0, 1, 1, 2, 5, 8, 9, 10, 10, 10, 10, 10, 10, 10, 9, 8, 5, 2, 1, 1, 0], dtype=float)
    p = make("infl", s21)
    assert abs(p.center_idx - 10) < 0.01 and abs(p.field_edge_idx("left") - 3.5) < 0.01
    assert abs(p.field_edge_idx("right") - 16.5) < 0.01 and abs(p.field_width_px - 13) < 0.01
    assert abs(make("infl", s20).center_idx - 9.5) < 0.01
    h = make("hill", sharp21, hill_window_ratio=0.2)
    assert abs(h.center_idx - 10) < 0.1 and abs(h.field_edge_idx("left") - 3.8) < 0.1
    assert abs(h.field_edge_idx("right") - 15.9) < 0.1 and abs(h.field_width_px - 12) < 0.1


def check_edge_profiles(g, make, only=None):
    """`make(kind, values, **kw)` -> an object with field_edge_idx / center_idx / field_width_px (+ optionally
    geometric_center_idx / cax_index).  Edges come out of BFGS on a cubic interpolant (gtol 1e-5) and, for "hill", a
    curve_fit on top: held to 1e-5 of the profile's extent (these optimisers are not bit-reproducible run to run)."""
    n = 0
    for tag, kind, values, kw in edge_profile_cases(g):
        if only is not None and not only(tag):
            continue
        if f"{tag}.error" in g:
            try:
                p = make(kind, values.copy(), **kw)
                p.field_edge_idx("left"), p.field_edge_idx("right")
            except (ValueError, IndexError, RuntimeError, TypeError):
                continue
            raise AssertionError(f"{tag}: the reference raised {g[tag + '.error']}")
        p = make(kind, values.copy(), **kw)
        want = g[tag]
        x = kw.get("x_values")
        extent = float(np.ptp(x)) if x is not None else float(len(values))
        got = [p.field_edge_idx("left"), p.field_edge_idx("right"), p.center_idx, p.field_width_px]
        assert np.allclose(got, want[:4], rtol=0, atol=1e-5 * extent), (tag, got, want[:4])
        if hasattr(p, "cax_index"):
            assert np.allclose([p.geometric_center_idx, p.cax_index], want[4:], rtol=0, atol=1e-9), tag
        n += 1
    return n


def check_catphan_volume(golden, dev, names=("a", "b")):
    """Volume-level CatPhan localisation (config #5's loop over slices) against the reference's own
    CatPhanBase.find_phantom_axis / find_origin_slice on synthetic volumes: which slices show the phantom, their ROI
    centroids (1e-9), the two axis fits (1e-9), the origin slice, the phantom roll from the air bubbles (1e-9 degrees)."""
    from pylinac_amd import ct

    g = golden("catphan_volume")
    for name in names:
        vol = torch.from_numpy(g[f"{name}.volume"]).to(dev)
        mmpp = float(g[f"{name}.mmpp"])
        fit_zx, fit_zy, roi = ct.find_phantom_axis_volume(vol, mmpp)
        assert np.array_equal(roi[:, 0] == 0, g[f"{name}.in_view"]), name
        seen = g[f"{name}.in_view"]
        assert np.allclose(roi[seen, 3:5], g[f"{name}.centroids"][seen], rtol=0, atol=1e-9), name
        assert np.allclose(fit_zx, g[f"{name}.fit_zx"], rtol=1e-9, atol=1e-9), (name, fit_zx)
        assert np.allclose(fit_zy, g[f"{name}.fit_zy"], rtol=1e-9, atol=1e-9), (name, fit_zy)
        origin = ct.find_origin_slice_volume(vol, mmpp, fit_zx, fit_zy, slice_thickness=2.5, roi=roi)
        assert origin == int(g[f"{name}.origin"]), (name, origin)
        roll = ct.find_phantom_roll_volume(vol, mmpp, origin, fit_zx)
        assert abs(roll - float(g[f"{name}.roll"])) < 1e-9, (name, roll, float(g[f"{name}.roll"]))


def check_profile_base_fields(g, dev=None):
    """ProfileBase.field_x_values / field_values / field_indices / resample_to through FWXMProfile against the reference
    (tests/golden/edge_profiles.npz) and the reference's literal known answers (tests_basic/core/test_profile.py:327-341:
    5 and 3 field values of the 9-point triangle for in_field_ratio 1 and 0.5)."""
    from pylinac_amd import profile

    tri = profile.FWXMProfile(np.array([0, 1, 2, 3, 4, 3, 2, 1, 0], dtype=float), fwxm_height=50)
    assert len(tri.field_values(in_field_ratio=1)) == 5 and len(tri.field_values(in_field_ratio=0.5)) == 3
    # the physical variants: the reference's known answers (tests_basic/core/test_profile.py:476-482, 540-550, 576-581, 641-646)
    long23 = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0], dtype=float)
    sharp21 = np.array([0, 1, 1, 2, 5, 8, 9, 10, 10, 10, 10, 10, 10, 10, 9, 8, 5, 2, 1, 1, 0], dtype=float)
    pf = profile.FWXMProfilePhysical(long23, fwxm_height=50, dpmm=2)
    assert pf.field_width_mm == pf.field_width_px / 2
    px = profile.FWXMProfilePhysical(long23, fwxm_height=50, dpmm=3).physical_x_values
    assert abs(min(px) - 1 / 6) < 1e-12 and abs(max(px) - (23 / 3 - 1 / 6)) < 1e-12
    pi_ = profile.InflectionDerivativeProfilePhysical(long23, dpmm=2)
    assert pi_.field_width_mm == pi_.field_width_px / 2
    ph = profile.HillProfilePhysical(sharp21, dpmm=2, hill_window_ratio=0.2)
    assert ph.field_width_mm == ph.field_width_px / 2
    nod = profile.FWXMProfilePhysical(long23, x_values=np.arange(23) * 0.25)
    assert nod.implicit_dpmm == 0.25 and np.array_equal(nod.physical_x_values, nod.x_values)
    for i in (0, 3, 7, 12, 19):
        p = profile.FWXMProfile(g[f"fx{i}.y"], x_values=g[f"fx{i}.x"], fwxm_height=50)
        for r in (1.0, 0.8, 0.5):
            assert np.array_equal(p.field_x_values(r), g[f"fx{i}.fwxm.field_x.{r}"]), (i, r)
            assert np.allclose(p.field_values(r), g[f"fx{i}.fwxm.field_v.{r}"], rtol=1e-12, atol=1e-12), (i, r)
            assert np.allclose(p.field_indices(r), g[f"fx{i}.fwxm.field_idx.{r}"], rtol=0, atol=1e-12), (i, r)
        ys = g[f"fx{i}.fwxm.x_at_y_in"]
        assert np.allclose(p.x_at_y(ys, "left"), g[f"fx{i}.fwxm.x_at_y_left"], rtol=1e-12, atol=1e-12), i
        assert np.allclose(p.x_at_y(ys, "right"), g[f"fx{i}.fwxm.x_at_y_right"], rtol=1e-12, atol=1e-12), i
        assert isinstance(p.x_at_y(float(ys[1]), "left"), float)
        tx = g[f"fx{i}.fwxm.resample_x"]
        q = p.resample_to(profile.FWXMProfile(np.ones(len(tx)), x_values=tx))
        assert isinstance(q, profile.FWXMProfile) and np.array_equal(q.x_values, tx)
        assert np.allclose(q.values, g[f"fx{i}.fwxm.resample_y"], rtol=1e-12, atol=1e-12), i
        try:
            p.resample_to(profile.FWXMProfile(np.ones(5), x_values=np.linspace(tx[0] * 10, tx[-1] * 10, 5)))
        except ValueError:
            pass
        else:
            raise AssertionError("extrapolation must raise")


def check_as_resampled(dev):
    """ProfileBase.as_resampled against scipy.ndimage.zoom itself (order 3, mode nearest, grid_mode False; scipy is
    present on the GPU box too) and the reference's known answers (tests_basic/core/test_profile.py:343-381, 483-500:
    lengths, preserved x range, type, similar maxima; an interpolation factor of 0.5)."""
    from scipy import ndimage

    from pylinac_amd import ops, profile

    rng = np.random.default_rng(3)
    for n in (5, 12, 23, 24, 63, 200):
        for f in (2, 10, 0.5, 3.3, 1.0, 7.25):
            v = rng.normal(size=(3, n)) * 100
            got = ops.zoom1d_cubic(torch.from_numpy(v).to(dev), f).cpu().numpy()
            want = np.stack([ndimage.zoom(r, zoom=f, order=3, grid_mode=False, mode="nearest") for r in v])
            assert got.shape == want.shape and np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max()), (n, f)
    long23 = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0], dtype=float)
    for cls, kw in ((profile.FWXMProfile, dict(fwxm_height=50)), (profile.InflectionDerivativeProfile, dict()),
                    (profile.FWXMProfilePhysical, dict(fwxm_height=50, dpmm=2))):
        p = cls(long23, **kw)
        for factor, length in ((2, 46), (10, 230), (0.5, 12)):
            if cls is profile.FWXMProfilePhysical:
                continue                                  # the physical classes resample by resolution (below)
            r = p.as_resampled(interpolation_factor=factor)
            assert len(r) == length and isinstance(r, cls) and r.x_values.max() == p.x_values.max()
            assert abs(r.values.max() - p.values.max()) < 0.1
            want = ndimage.zoom(long23, zoom=factor, order=3, grid_mode=False, mode="nearest")
            assert np.allclose(r.values, want, rtol=1e-12, atol=1e-12)
    # grid mode + the physical variants: the reference's known answers (tests_basic/core/test_profile.py:483-538, 583-638)
    v = rng.normal(size=(2, 41)) * 10
    for f in (0.5, 2.0, 10.0, 3.7):
        got = ops.zoom1d_cubic(torch.from_numpy(v).to(dev), f, grid_mode=True).cpu().numpy()
        want = np.stack([ndimage.zoom(r, zoom=f, order=3, grid_mode=True, mode="nearest") for r in v])
        assert got.shape == want.shape and np.allclose(got, want, rtol=1e-12, atol=1e-11), f
    for cls, kw in ((profile.FWXMProfilePhysical, dict(fwxm_height=50)), (profile.InflectionDerivativeProfilePhysical, dict()),
                    (profile.HillProfilePhysical, dict(hill_window_ratio=0.2))):
        same = cls(long23, dpmm=2, **kw).as_resampled(interpolation_resolution_mm=0.5)
        assert len(same) == 23 and isinstance(same, cls) and same.x_values.max() == 22
        p1 = cls(long23, dpmm=1, **kw)
        r10 = p1.as_resampled(interpolation_resolution_mm=0.1)
        assert len(r10) == 230 and isinstance(r10, cls) and r10.x_values.max() == p1.x_values.max() + 0.45
        assert r10.dpmm == 10 and abs(r10.x_values[0] + 0.45) < 0.01 and abs(r10.values.max() - 10) < 0.1
        want = ndimage.zoom(long23, zoom=10, order=3, grid_mode=True, mode="nearest")
        assert np.allclose(r10.values, want, rtol=1e-12, atol=1e-12)
        # resampling a resampled profile again must not use grid mode (:620-638)
        r100 = p1.as_resampled(interpolation_resolution_mm=0.01)
        r100_2 = r10.as_resampled(interpolation_resolution_mm=0.01, grid=False)
        assert len(r100.values) == len(r100_2.values)
    simple = profile.FWXMProfilePhysical(long23, fwxm_height=50, dpmm=2).as_simple_profile()
    assert type(simple) is profile.FWXMProfile and np.allclose(simple.x_values, np.arange(23) / 2 + 0.25)
    ints = profile.FWXMProfile((long23 * 1000).astype(np.int32), fwxm_height=50).as_resampled(3)
    want = ndimage.zoom((long23 * 1000).astype(np.int32), zoom=3, order=3, grid_mode=False, mode="nearest")
    assert ints.values.dtype == want.dtype and np.array_equal(ints.values, want)


def check_bit_invert_and_convert_to_dtype():
    """array_utils.bit_invert / convert_to_dtype: the reference's literal known answers
    (tests_basic/core/test_array_utils.py:152-173, 215-243) and numpy's own expressions on random arrays."""
    from pylinac_amd import array_utils as au

    assert np.array_equal(au.bit_invert(np.array([0, 10], dtype=np.uint8)), [255, 245])
    assert np.array_equal(au.bit_invert(np.array([0, 10], dtype=np.uint16)), [65535, 65525])
    assert np.array_equal(au.bit_invert(np.array([0, 10], dtype=np.int8)), [-1, -11])
    try:
        au.bit_invert(np.array([0, 10], dtype=float))
    except ValueError as exc:
        assert "could not be safely" in str(exc)
    else:
        raise AssertionError("float input must raise")
    c = au.convert_to_dtype(np.array([5, 6, 7], dtype=np.uint8), dtype=np.uint16)
    assert np.array_equal(c, [1285, 1542, 1799]) and c.dtype == np.uint16
    c = au.convert_to_dtype(np.array([0, 100, 1000, 10000, 65535], dtype=np.uint16), dtype=np.uint8)
    assert np.array_equal(c, [0, 1, 4, 39, 255]) and c.dtype == np.uint8
    c = au.convert_to_dtype(np.array([0, 255], dtype=np.uint8), dtype=np.int8)
    assert np.array_equal(c, [-128, 127]) and c.dtype == np.int8
    c = au.convert_to_dtype(np.array([0, 255.2], dtype=float), dtype=np.uint16)
    assert np.array_equal(c, [0, 65535]) and c.dtype == np.uint16
    rng = np.random.default_rng(2)
    for dt in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.int64, np.uint64):
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, (13, 17), dtype=dt, endpoint=True)
        a.ravel()[:4] = [info.min, info.max, info.max - 1, info.min + 1]      # the extremes (ADVICE: 64-bit, 2^63)
        got = au.bit_invert(a)
        assert got.dtype == a.dtype and np.array_equal(got, np.invert(a)), dt
    assert np.array_equal(au.bit_invert(np.array([True, False])), [False, True])
    # 64-bit invert()/ground(): values beyond 2^53 cannot pass through the float64 extrema -> refused, not wrong
    big = np.array([2**60, 5, 7], dtype=np.int64)
    for fn in (au.invert, au.ground):
        try:
            fn(big)
        except TypeError:
            pass
        else:
            raise AssertionError("values beyond 2**53 must be refused")
    assert np.array_equal(au.invert(np.array([1, 5, 9], dtype=np.int64)), [9, 5, 1])
    # ground() with a non-integral value promotes an integer array to float64, like numpy
    g16 = au.ground(np.array([3, 5, 9], dtype=np.uint16), 0.5)
    assert g16.dtype == np.float64 and np.array_equal(g16, [0.5, 2.5, 6.5])
    assert au.ground(np.array([3, 5, 9], dtype=np.uint16), 2).dtype == np.uint16
    # find_peaks(max_number=0) is the reference's `[::-1][:0]`: nothing; a negative count is refused
    from pylinac_amd import profile as pr
    idx, props = pr.find_peaks(np.array([0, 1, 0, 2, 0, 3, 0], dtype=float), max_number=0)
    assert len(idx) == 0 and all(len(v) == 0 for v in props.values())
    try:
        pr.find_peaks(np.array([0, 1, 0], dtype=float), max_number=-1)
    except ValueError:
        pass
    else:
        raise AssertionError("negative max_number must raise")
    a16 = rng.integers(0, 65535, (9, 21)).astype(np.uint16)
    for new in (np.uint8, np.int16, np.uint16):
        ninfo = np.iinfo(new)
        with np.errstate(all="ignore"):
            want = np.array(a16.astype(float) / 65535 * (ninfo.max - ninfo.min) - ninfo.max - 1, dtype=new)
        got = au.convert_to_dtype(a16, new)
        assert got.dtype == want.dtype and np.array_equal(got, want), new


def check_rotate(golden, dev):
    """BaseImage.rotate (pylinac/core/image.py:780-783) == skimage.transform.rotate(array, angle, mode="edge") of
    scikit-image 0.18.3 (tests/golden/rotate.npz, generated by tests/golden/skimage_rotate_py39.py).  Three statements:
    (1) the kernel on skimage's own captured inverse map and skimage's float conversion is BIT-EXACT; (2) the host builds
    that map to within 2 ulp of its largest entry (the 3x3 products run through whichever BLAS numpy links); (3) the public
    ``ArrayImage.rotate`` therefore agrees to 1e-12 of the value range, and exactly for the right angles' dtype / shape."""
    import torch

    from pylinac_amd import image as im
    from pylinac_amd import ops

    g = golden("rotate")
    angles = g["angles"]
    for name in g["names"]:
        name = str(name)
        a = g[name + ".in"]
        k = 0
        while f"{name}.{k}.out" in g:
            want, m = g[f"{name}.{k}.out"], g[f"{name}.{k}.m"]
            # (1) same conversion, same map -> same bits
            if a.dtype.kind == "u":
                x = np.multiply(a, 1.0 / np.iinfo(a.dtype).max, dtype=np.float64)
            elif a.dtype.kind == "i":
                x = np.add(a, 0.5, dtype=np.float64)
                x *= 2 / (int(np.iinfo(a.dtype).max) - int(np.iinfo(a.dtype).min))
            else:
                x = a.astype(np.float64) if a.dtype.kind == "b" else a
            got = ops.warp_affine(torch.from_numpy(np.ascontiguousarray(x)).to(dev), m, order=0 if a.dtype.kind == "b" else 1)[0].cpu().numpy()
            assert got.dtype == want.dtype and got.shape == want.shape, (name, k)
            assert np.array_equal(got, want), (name, k, float(np.abs(got - want).max()))
            # (2) the host's map
            mine = ops.rotation_matrix(a.shape[0], a.shape[1], float(angles[k]))
            assert np.abs(mine - m).max() <= 2 * np.spacing(np.abs(m).max()), (name, k)
            # (3) the public method
            img = im.ArrayImage(a.copy())
            img.rotate(float(angles[k]))
            assert img.array.dtype == want.dtype and img.array.shape == want.shape
            span = float(want.max() - want.min()) or 1.0
            assert np.abs(img.array - want).max() <= 1e-12 * span, (name, k, float(np.abs(img.array - want).max()))
            k += 1
    # contract: what is not offered fails loudly; clip=False leaves the interpolated values alone
    a = g["f64.in"]
    for kw in (dict(resize=True), dict(order=3), dict(mode="constant")):
        try:
            im.rotate_array(a, 10.0, **kw)
        except NotImplementedError:
            pass
        else:
            raise AssertionError(kw)
    assert np.array_equal(im.rotate_array(a, 33.3, clip=False), im.rotate_array(a, 33.3))     # bilinear stays in range
    # batched device frames: every frame is rotated like the single image
    t = torch.from_numpy(np.stack([a, a[::-1].copy(), a * 2])).to(dev)
    r = im.rotate_array(t, -12.5)
    for i in range(3):
        assert np.array_equal(r[i].cpu().numpy(), im.rotate_array(t[i].cpu().numpy(), -12.5))


def check_large_rois(dev):
    """ROIs beyond the LDS buffer (DiskROI of radius > ~72 px such as the ACR large uniformity ROI, large
    rectangles) go through the streaming path: same statistics as numpy on the same pixel set.  And the out-of-frame
    decision comes from the pixels a disk actually selects: a disk whose box reaches row -1 without selecting anything
    there (cy - r == -1: the boundary is excluded) is inside, as in the reference (pylinac/core/roi.py:134-138)."""
    from oracle import pylinac_oracle as o
    from pylinac_amd import roi

    rng = np.random.default_rng(8)
    arr = (rng.normal(1000, 50, (400, 460)) + np.add.outer(np.arange(400), np.arange(460)) * 0.5)
    frames = {"f64": arr, "i16": arr.astype(np.int16), "u16": arr.astype(np.uint16)}
    disks = np.array([[230.3, 200.7, 120.0], [100.0, 90.0, 72.5], [300.5, 250.5, 149.0], [50.0, 11.0, 12.0],
                      [11.0, 60.0, 12.0], [200.0, 200.0, 5.0]])
    for name, a in frames.items():
        out, st = roi.disk_roi_stats_batch(torch.from_numpy(a).to(dev)[None], disks[:, :2], disks[:, 2])
        assert int(st.abs().sum()) == 0, (name, st)
        got = out[0].cpu().numpy()
        want = np.array([o.disk_roi_stats(a, cx, cy, r) for cx, cy, r in disks])
        assert want[0, 0] > 16384 and want[2, 0] > 4 * 16384       # beyond the LDS buffer / beyond the gather box limit
        assert np.array_equal(got[:, [0, 3, 4, 5]], want[:, [0, 3, 4, 5]]), name
        assert np.allclose(got[:, 1:3], want[:, 1:3], rtol=1e-12, atol=0), name
    # one pixel further and the disk selects row / column -1: reported like before
    _, st = roi.disk_roi_stats_batch(torch.from_numpy(arr).to(dev)[None], [[50.0, 10.5], [10.5, 60.0]], 12.0)
    assert st.cpu().numpy().tolist() == [[1, 1]]
    d = roi.DiskROI(frames["i16"], radius=120.0, center=(230.3, 200.7))
    w = o.disk_roi_stats(frames["i16"], 230.3, 200.7, 120.0)
    assert (d.pixel_value, d.min, d.max) == (w[5], w[3], w[4]) and abs(d.mean - w[1]) < 1e-9 and abs(d.std - w[2]) < 1e-9
    # large windows and a large rotated rectangle
    boxes = np.array([[0, 400, 0, 460], [10, 390, 20, 300], [100, 228, 100, 229]], dtype=float)
    outr, st = roi.rectangle_stats_batch(torch.from_numpy(arr).to(dev)[None], boxes)
    assert int(st.abs().sum()) == 0
    for k, (r0, r1, c0, c1) in enumerate(boxes.astype(int)):
        v = arr[r0:r1, c0:c1]
        got = outr[0, k].cpu().numpy()
        assert got[0] == v.size and got[3] == v.min() and got[4] == v.max() and got[5] == np.median(v)
        assert abs(got[1] - v.mean()) <= 1e-12 * abs(v.mean()) and abs(got[2] - v.std()) <= 1e-12 * v.std()
    big = roi.RectangleROI(frames["i16"], width=300, height=180, center=(230, 200), rotation=30)
    small_sum = roi.RectangleROI(frames["i16"], width=300, height=180, center=(230, 200), rotation=0)
    v = frames["i16"][110:290, 80:380]                      # rotation 0: pixels_flat's -1 corners make it == pixel_array
    assert small_sum._s()[0] == v.size and small_sum.min == v.min() and small_sum.max == v.max()
    assert abs(small_sum.mean - v.mean()) < 1e-9 and abs(small_sum.std - v.std()) < 1e-9
    n_big = big._s()[0]
    assert abs(n_big - 300 * 180) < 0.02 * 300 * 180 and big.min >= frames["i16"].min() and big.max <= frames["i16"].max()


# ---- round 3: the fused CatPhan localisation kernels (slice_regions.hip, edge_field.hip, circle.hip) -------------------
def _regions_reference(bw, ext, fill, max_labels):
    """clear_border -> binary_fill_holes -> label(8-connected) -> raw regionprops sums with scipy (scikit-image's label /
    clear_border number and select components the same way; ndimage.label numbers in raster order of the first pixel)."""
    from scipy import ndimage

    bw = bw.astype(bool).copy()
    s8 = np.ones((3, 3))
    if ext > 0:
        lab, _ = ndimage.label(bw, s8)
        band = np.zeros_like(bw)
        band[:ext] = band[-ext:] = True
        band[:, :ext] = band[:, -ext:] = True
        bw[np.isin(lab, np.unique(lab[band & bw])) & bw] = False
    if fill:
        bw = ndimage.binary_fill_holes(bw)
    lab, n = ndimage.label(bw, s8)
    tab = np.zeros((max_labels, 7))
    for k in range(1, min(n, max_labels) + 1):
        rr, cc = np.nonzero(lab == k)
        tab[k - 1] = [len(rr), rr.min(), cc.min(), rr.max() + 1, cc.max() + 1, rr.sum(), cc.sum()]
    return tab, n, bw.astype(np.uint8)


def check_mask_regions(dev, shapes=((64, 64), (70, 130), (33, 65), (17, 5), (96, 192), (1, 1), (40, 64)), seed=1):
    """pl_mask_regions against scipy on random speckle, smooth blobs, full and empty masks: final mask, label count and the
    region table are exact; a frame with more runs than the LDS list holds must say so (status 1) and nothing else may."""
    import torch
    from scipy import ndimage

    from pylinac_amd import ops

    rng = np.random.default_rng(seed)
    checked = overflowed = 0
    for h, w in shapes:
        speckle = rng.random((2, h, w)) < rng.choice([0.3, 0.5, 0.7])
        blobs = ndimage.gaussian_filter(rng.random((3, h, w)), (0, 2, 2)) > 0.5
        arr = np.concatenate([speckle, blobs, np.ones((1, h, w), bool), np.zeros((1, h, w), bool)]).astype(np.uint8)
        for ext, fill in ((0, False), (1, False), (3, True), (0, True), (4, True)):
            if 2 * ext > min(h, w):
                continue
            tab, cnt, st, om = ops.mask_regions(torch.from_numpy(arr).to(dev), None, ext, fill, 64, return_mask=True)
            tab, cnt, st, om = tab.cpu().numpy(), cnt.cpu().numpy(), st.cpu().numpy(), om.cpu().numpy()
            for i in range(arr.shape[0]):
                if st[i]:
                    runs = int((np.diff(np.pad(arr[i].astype(np.int8), ((0, 0), (1, 1))), axis=1) == 1).sum())
                    assert st[i] == 1 and 2 * runs + h > 3072, (h, w, i, runs)      # only genuinely crowded frames
                    overflowed += 1
                    continue
                rt, rn, rbw = _regions_reference(arr[i], ext, fill, 64)
                assert cnt[i] == rn, (h, w, ext, fill, i, cnt[i], rn)
                assert np.array_equal(om[i], rbw), (h, w, ext, fill, i)
                assert np.array_equal(tab[i], rt), (h, w, ext, fill, i)
                checked += 1
    # round 6: the clear_border labelling is skipped when the border band holds no foreground pixel -- an interior blob with
    # an empty band, then one arm of it reaching exactly the first / last column and row inside and outside the band (the last
    # columns of a row live in the top bits of its last word, or across two words when w % 64 < ext)
    for h, w in ((40, 64), (70, 130), (33, 65), (96, 192), (48, 67)):
        for ext in (1, 3, 4):
            base = np.zeros((h, w), np.uint8)
            base[h // 3:2 * h // 3, w // 3:2 * w // 3] = 1
            base[h // 2 - 2:h // 2 + 2, w // 2 - 2:w // 2 + 2] = 0                    # a hole
            cases = [base]
            r, c = h // 2, w // 2
            for lo_c in (0, ext - 1, ext):                                           # left arm ends inside / outside the band
                m = base.copy(); m[r, lo_c:w // 3] = 1; cases.append(m)
            for hi_c in (w - 1, w - ext, w - ext - 1):
                m = base.copy(); m[r, 2 * w // 3:hi_c + 1] = 1; cases.append(m)
            for lo_r in (0, ext - 1, ext):
                m = base.copy(); m[lo_r:h // 3, c] = 1; cases.append(m)
            for hi_r in (h - 1, h - ext, h - ext - 1):
                m = base.copy(); m[2 * h // 3:hi_r + 1, c] = 1; cases.append(m)
            m = base.copy(); m[0, 0] = 1; cases.append(m)                             # a lone corner pixel
            m = base.copy(); m[h - 1, w - 1] = 1; cases.append(m)
            arr = np.stack(cases)
            tab, cnt, st, om = ops.mask_regions(torch.from_numpy(arr).to(dev), None, ext, True, 64, return_mask=True)
            assert not st.any()
            for i in range(len(cases)):
                rt, rn, rbw = _regions_reference(arr[i], ext, True, 64)
                assert int(cnt[i]) == rn and np.array_equal(om[i].cpu().numpy(), rbw) and np.array_equal(tab[i].cpu().numpy(), rt), (h, w, ext, i)
                checked += 1
    # float64 frames with per-frame thresholds (strictly greater; NaN is background)
    e = rng.random((2, 50, 70))
    e[0, 3, 3] = np.nan
    thr = np.array([0.5, 0.7])
    tab, cnt, st = ops.mask_regions(torch.from_numpy(e).to(dev), torch.from_numpy(thr).to(dev), 2, True, 64)
    for i in range(2):
        with np.errstate(invalid="ignore"):
            rt, rn, _ = _regions_reference(e[i] > thr[i], 2, True, 64)
        assert int(cnt[i]) == rn and np.array_equal(tab[i].cpu().numpy(), rt)
    return checked, overflowed


def check_scharr_gaussian(dev, shapes=((2, 70, 130, np.int16), (1, 33, 65, np.uint16), (2, 32, 64, np.int16), (1, 5, 7, np.int16),
                                       (1, 100, 9, np.uint16), (1, 64, 200, np.int16)), sigmas=(1, 0.5, 2)):
    """pl_scharr_gaussian == pl_scharr -> pl_gaussian2d_mode('nearest') -> pl_minmax / pl_minmax_masked, bit for bit (those
    are pinned to scikit-image 0.18.3 / scipy by the golden tests), on shapes that are not multiples of the 32 x 64 tile."""
    import torch

    from pylinac_amd import ops

    rng = np.random.default_rng(3)
    for n, h, w, dt in shapes:
        a = rng.integers(-1000 if dt == np.int16 else 0, 3000, (n, h, w)).astype(dt)
        a[:, h // 4:h // 2, w // 4:w // 2] += 500
        m = (rng.random((h, w)) < 0.6).astype(np.uint8)
        m[0, 0] = 1
        x, mt = torch.from_numpy(a).to(dev), torch.from_numpy(m).to(dev)
        raw = ops.scharr(x)
        for sigma in sigmas:
            e2 = ops.gaussian_filter_mode(raw, sigma, "nearest")
            e, rm, lo, hi = ops.scharr_gaussian(x, sigma, mt)
            assert torch.equal(e, e2), (n, h, w, sigma)
            assert torch.equal(rm, ops.minmax(raw)[1])
            lo2, hi2 = ops.minmax_masked(e2, mt)
            assert torch.equal(lo, lo2) and torch.equal(hi, hi2)
            _, _, lo, hi = ops.scharr_gaussian(x, sigma, None)
            lo2, hi2 = ops.minmax(e2)
            assert torch.equal(lo, lo2) and torch.equal(hi, hi2)
            # the streaming entry point itself: float32 plane = RN of the float64 one, row spans = the same selection as a
            # byte mask whose rows are single runs, extrema-only call
            c0 = rng.integers(0, w, h)
            c1 = np.minimum(c0 + rng.integers(0, w, h), w)
            ms = np.zeros((h, w), np.uint8)
            for r in range(h):
                ms[r, c0[r]:c1[r]] = 1
            spans = torch.from_numpy(np.stack([c0, c1], 1).astype(np.int32)).to(dev)
            p32, rm3, lo3, hi3 = ops.edge_plane(x, sigma, spans=spans)
            assert p32.dtype == torch.float32 and torch.equal(p32, e2.to(torch.float32)) and torch.equal(rm3, rm)
            _, _, lo4, hi4 = ops.scharr_gaussian(x, sigma, torch.from_numpy(ms).to(dev))
            assert torch.equal(lo3, lo4) and torch.equal(hi3, hi4)
            none, rm5, lo5, hi5 = ops.edge_plane(x, sigma, spans=spans, want_plane=False)
            assert none is None and torch.equal(rm5, rm) and torch.equal(lo5, lo4) and torch.equal(hi5, hi4)


PF_ORIENT_CASES = (("lr", "lr", "LEFT_RIGHT", False), ("lr_sep", "lr", "LEFT_RIGHT", True), ("ud_sep", "ud_sep", "UP_DOWN", True))


def check_pf_orientation_oracle(g):
    """oracle.pf_measure(orientation=, separate_leaves=) against the reference's OWN PicketFence.analyze() on a LEFT_RIGHT
    frame, the same frame with separate_leaves, and an UP_DOWN frame with separate_leaves
    (tests/golden/make_pf_orient_golden.py): spacing, picket indices and every position the reference kept, bit for bit."""
    from oracle import pylinac_oracle as o

    for name, frame_key, orient, sep in PF_ORIENT_CASES:
        raw, dpmm = g[f"{frame_key}.cropped"], float(g[f"{frame_key}.dpmm"])
        r = o.pf_measure(o.normalize(o.ground(raw)), dpmm, orientation=orient, separate_leaves=sep)
        assert r["spacing"] == float(g[f"{name}.spacing"])
        idx = {n: i for i, (n, c, w) in enumerate(r["leaves"])}
        meas = g[f"{name}.meas"]
        assert len(meas) > 400
        for row in meas:
            leaf, picket, approx = int(row[0]), int(row[1]), row[2]
            assert r["peak_idxs"][picket] == approx
            if sep:
                assert (r["left"][idx[leaf], picket], r["right"][idx[leaf], picket]) == (row[3], row[4])
            else:
                assert r["position"][idx[leaf], picket] == row[3]


def check_pf_orientation_device(g, dev):
    """picketfence.analyze_batch(orientation=, separate_leaves=) == the oracle on every window (NaN pattern included) and ==
    the reference's own analyze() on every window it kept."""
    import torch

    from oracle import pylinac_oracle as o
    from pylinac_amd import picketfence as ppf

    for name, frame_key, orient, sep in PF_ORIENT_CASES:
        raw, dpmm = g[f"{frame_key}.cropped"], float(g[f"{frame_key}.dpmm"])
        res = ppf.analyze_batch(torch.from_numpy(np.ascontiguousarray(raw)[None]).to(dev), dpmm, orientation=orient,
                                separate_leaves=sep)
        ref = o.pf_measure(o.normalize(o.ground(raw)), dpmm, orientation=orient, separate_leaves=sep)
        P = len(ref["peak_idxs"])
        assert int(res.picket_count[0]) == P and float(res.spacing[0]) == ref["spacing"] == float(g[f"{name}.spacing"])
        assert np.array_equal(res.picket_idx[0, :P].cpu().numpy(), ref["peak_idxs"])
        assert res.leaf_nums == [n for n, _, _ in ref["leaves"]]
        pairs = [(res.position, ref["position"])] + ([(res.left, ref["left"]), (res.right, ref["right"])] if sep else [])
        for got_t, want in pairs:
            got = got_t[0, :, :P].cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)), name
            assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(got)]), name
        idx = {n: i for i, n in enumerate(res.leaf_nums)}
        for row in g[f"{name}.meas"]:                      # what the reference itself measured
            leaf, picket = int(row[0]), int(row[1])
            if sep:
                assert (float(res.left[0, idx[leaf], picket]), float(res.right[0, idx[leaf], picket])) == (row[3], row[4])
            else:
                assert float(res.position[0, idx[leaf], picket]) == row[3]


def check_ground_promotion():
    """array_utils.ground == ``array - array.min() + value`` in numpy's dtype AND numpy's arithmetic for every kind of value:
    a narrow numpy float scalar is added in the narrow float type (no float64 detour), and the integer subtraction
    wraps like numpy's before the promotion (an int16 frame spanning more than 32 767 levels)."""
    from pylinac_amd import array_utils as au

    arrays = (np.array([[5, 9, 7], [6, 5, 8]], dtype=np.uint16), np.array([[5, 9, 7], [6, 65535, 8]], dtype=np.uint16),
              np.array([[-32768, 32767, 5]], dtype=np.int16), np.array([[3, 250]], dtype=np.uint8))
    for a in arrays:
        for value in (0, 3, 0.0, 1.0, 2.5, 0.1, np.float64(1.0), np.float64(0.1), np.float32(1.5), np.float32(0.1), np.float16(0.3),
                      float("inf")):
            with np.errstate(over="ignore"):
                want = a - a.min() + value
            got = au.ground(a, value)
            assert got.dtype == want.dtype and np.array_equal(got, want), (a.dtype, value, got.dtype, want.dtype)


def check_edge_otsu(dev, shapes=((2, 70, 130, np.int16), (1, 33, 65, np.uint16), (1, 100, 9, np.uint16), (2, 64, 200, np.int16)),
                    sigmas=(1, 2)):
    """pl_edge_otsu (one launch; float32 plane with exact recomputation of the undecided pixels, and float64 plane) ==
    pl_linspace_edges -> pl_hist_uniform -> pl_otsu_from_counts on the float64 plane (the golden-pinned round 1-3 path), on
    row-span and byte-mask selections, constant and empty selections included.  -> how many pixels took the exact path is not
    observable from outside; the float32 case is additionally forced through it by a plane whose values sit ON bin edges."""
    import torch

    from pylinac_amd import ops

    rng = np.random.default_rng(17)
    n_exact = 0
    for n, h, w, dt in shapes:
        a = rng.integers(-1000 if dt == np.int16 else 0, 3000, (n, h, w)).astype(dt)
        a[:, h // 4:h // 2, w // 4:w // 2] += 700
        a[0, : h // 8] = 5                                        # a flat band: edge value exactly 0 there
        x = torch.from_numpy(a).to(dev)
        c0 = rng.integers(0, w // 2, h)
        c1 = np.minimum(c0 + rng.integers(1, w, h), w)
        ms = np.zeros((h, w), np.uint8)
        for r in range(h):
            ms[r, c0[r]:c1[r]] = 1
        spans = torch.from_numpy(np.stack([c0, c1], 1).astype(np.int32)).to(dev)
        mt = torch.from_numpy(ms).to(dev)
        for sigma in sigmas:
            e64, _, lo, hi = ops.edge_plane(x, sigma, spans=spans, dtype=torch.float64)
            want_thr, want_raw = ops.otsu_float_masked(e64, mt, scale=0.8, lohi=(lo, hi))
            p32 = e64.to(torch.float32)
            for plane, kw in ((p32, dict(frames=x, sigma=sigma)), (e64, {})):
                for sel in (dict(spans=spans), dict(mask=mt)):
                    thr, raw, work = ops.edge_otsu(plane, lo, hi, scale=0.8, return_work=True, **kw, **sel)
                    assert torch.equal(raw, want_raw) and torch.equal(thr, want_thr), (n, h, w, sigma, plane.dtype, list(sel))
                    # round 6 (float32 planes are binned in the bit domain): all 256 counts, not only the threshold they give
                    lin0 = torch.empty((n, 257), dtype=torch.float64, device=e64.device)
                    from pylinac_amd import _lib as _l
                    ops.check(_l.load().pl_linspace_edges(lo.data_ptr(), hi.data_ptr(), 256, n, lin0.data_ptr(), ops._stream()), "edges")
                    assert torch.equal(work[:, :256], ops.hist_uniform(e64, lin0, mt)), (n, h, w, sigma, plane.dtype, list(sel))
            # a NARROW range (np.histogram drops what lies outside): the bins shrink to a few float32 steps, so the float32
            # plane cannot decide most pixels inside and the exact recomputation carries the histogram
            sel_vals = e64[0][mt.bool()]
            mid = sel_vals.sort().values[sel_vals.numel() // 2]
            nlo = (mid * (1 - 3e-5)).reshape(1).expand(n).contiguous()
            nhi = (mid * (1 + 3e-5)).reshape(1).expand(n).contiguous()
            want_thr, want_raw = ops.otsu_float_masked(e64, mt, scale=0.8, lohi=(nlo, nhi))
            thr, raw, work = ops.edge_otsu(p32, nlo, nhi, frames=x, sigma=sigma, spans=spans, scale=0.8, return_work=True)
            assert torch.equal(raw, want_raw) and torch.equal(thr, want_thr)
            lin = torch.empty((n, 257), dtype=torch.float64, device=e64.device)
            from pylinac_amd import _lib
            ops.check(_lib.load().pl_linspace_edges(nlo.data_ptr(), nhi.data_ptr(), 256, n, lin.data_ptr(), ops._stream()), "edges")
            assert torch.equal(work[:, :256], ops.hist_uniform(e64, lin, mt))
            n_exact += int(work[:, 257].sum())
    assert n_exact > 0, "the exact path was never taken"
    # constant selection -> the value; empty selection -> NaN (float64 plane: no recomputation possible or needed)
    flat = torch.full((1, 40, 70), 3.25, dtype=torch.float64, device=dev)
    sp = torch.from_numpy(np.tile(np.array([[5, 60]], np.int32), (40, 1))).to(dev)
    v = torch.tensor([3.25], dtype=torch.float64, device=dev)
    thr, raw = ops.edge_otsu(flat, v, v, spans=sp, scale=0.8)
    assert float(raw[0]) == 3.25 and float(thr[0]) == 3.25 * 0.8
    inf = torch.tensor([float("inf")], dtype=torch.float64, device=dev)
    thr, raw = ops.edge_otsu(flat, inf, -inf, spans=sp, scale=0.8)
    assert np.isnan(float(raw[0]))


def check_circle_profile_combined(dev, n_volumes=2, spv=9, h=96, w=112):
    """The per-tap maximum over the +-k slices == the profile of pl_combine_slices' planes: every slice of two volumes
    (wrapping first slices, clamped last ones included), a subset through the slice index, k = 0 .. 3."""
    import torch

    from pylinac_amd import ops

    rng = np.random.default_rng(11)
    vol = torch.from_numpy(rng.integers(-1000, 3000, (n_volumes * spv, h, w)).astype(np.int16)).to(dev)
    n = n_volumes * spv
    cx = w / 2 + rng.uniform(-3, 3, n)
    cy = h / 2 + rng.uniform(-3, 3, n)
    radii = np.linspace(30.0, 34.0, 5)
    size = np.pi * radii.max() * 2 * 2
    for k in (0, 1, 3):
        combined = ops.combine_slices(vol, k, "max", spv)
        want = ops.circle_profile(combined, cx, cy, radii, size, np.pi, True, 5.0)
        got = ops.circle_profile(vol, cx, cy, radii, size, np.pi, True, 5.0, combine=(np.arange(n), spv, k))
        assert torch.equal(got, want), k
        sub = np.array([0, spv - 1, spv, n - 1, 4])
        got = ops.circle_profile(vol, cx[sub], cy[sub], radii, size, np.pi, True, 5.0, combine=(sub, spv, k))
        assert torch.equal(got, want[torch.from_numpy(sub).to(dev)]), k


def check_circle_profile_ring(dev, n_volumes=2, spv=5, h=120, w=136, light=False):
    """pl_circle_profile_ring (the annulus staged in LDS) against pl_circle_profile_combined_ex (scattered gathers), samples AND
    margins bit for bit: rings inside the frame, cut by every border, centred outside it, a NaN centre; uint8 / int16 / uint16
    / int32 slices; k = 0 .. 3; a promise about the radii that is too narrow on either side (the taps outside it must come
    from the slices themselves) and one so wide that the box exceeds the LDS (the entry point forwards)."""
    import ctypes as C

    import torch

    from pylinac_amd import _lib, ops

    lib = _lib.load()
    rng = np.random.default_rng(23)
    n = n_volumes * spv
    centres = np.stack([w / 2 + rng.uniform(-4, 4, n), h / 2 + rng.uniform(-4, 4, n)], 1)
    centres[1] = (6.3, 50.2)               # cut by the left border
    centres[2] = (w - 4.6, h - 3.1)        # the bottom-right corner
    centres[3] = (-20.5, -11.25)           # outside the frame
    centres[4] = (w / 2, 2.0)
    centres[5] = (np.nan, 40.0)
    n_checked = 0
    cases = ((np.int16, -1000, 3000), (np.uint16, 0, 65535), (np.uint8, 0, 255), (np.int32, -70000, 70000))
    for dtype, lo, hi in (cases[:2] if light else cases):
        vol = torch.from_numpy(rng.integers(lo, hi, (n, h, w)).astype(dtype)).to(dev)
        for radii in (np.linspace(30.0, 34.0, 7), np.linspace(3.0, 9.0, 20), np.array([0.0, 1.5, np.nan, 47.25]),
                      np.linspace(20.0, 26.0, 41)):      # (41 radii: beyond the 32 a lane group keeps in registers)
            size = np.pi * np.nanmax(radii) * 2 * 2
            d_cos, d_sin, nsamp = ops._circle_tables(size, np.pi, True, dev)
            r = torch.from_numpy(np.broadcast_to(radii[None, :], (n, len(radii))).copy()).to(dev)
            cx = torch.from_numpy(centres[:, 0].copy()).to(dev)
            cy = torch.from_numpy(centres[:, 1].copy()).to(dev)
            sidx = torch.arange(n, dtype=torch.int64, device=dev)
            for k in (((1, 3) if light else (0, 1, 2, 3)) if dtype == np.int16 else (1,)):
                def run(entry, *promise):
                    out = torch.full((n, nsamp), -7.0, dtype=torch.float64, device=dev)
                    mrg = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
                    args = [vol.data_ptr(), ops._dt(vol), n, h, w, sidx.data_ptr(), n, spv, k, d_cos.data_ptr(), d_sin.data_ptr(),
                            nsamp, r.data_ptr(), len(radii), cx.data_ptr(), cy.data_ptr(), float(len(radii))]
                    rc = getattr(lib, entry)(*args, *promise, out.data_ptr(), mrg.data_ptr(), ops._stream())
                    _lib.check(rc, entry)
                    return out.cpu().numpy(), mrg.cpu().numpy()

                want, want_m = run("pl_circle_profile_combined_ex")
                r_min, r_max = float(np.nanmin(radii)), float(np.nanmax(radii))      # (a NaN radius: its taps read 0)
                for promise in ((r_min, r_max),                                     # honest
                                (r_min + 2.0, r_max - 2.0) if r_max - r_min > 4 and len(radii) > 4 else (0.0, 0.5),
                                (0.0, 400.0)):                                      # box > 150 KB: forwarded
                    got, got_m = run("pl_circle_profile_ring", *promise)
                    assert np.array_equal(got, want, equal_nan=True), (dtype, radii[:2], k, promise)
                    assert np.array_equal(got_m, want_m, equal_nan=True), (dtype, radii[:2], k, promise, got_m, want_m)
                    n_checked += 1
    return n_checked


def check_phantom_roi_fused_vs_separate(dev, slices=(0, 24, 44, 79)):
    """ct.phantom_roi_batch (three launches: pl_edge_plane float32, pl_edge_otsu, pl_edge_regions with the ROI chosen on the
    device) == the same table built from the separate entry points (get_regions_batch + the host selection: the golden-pinned
    round-1/2 path) on synthetic CatPhan slices; every intermediate is compared too."""
    import torch

    from pylinac_amd import ct, ops
    from pylinac_amd.synthetic import catphan_volume

    vol = catphan_volume(seed=4000, n_slices=80)
    x = torch.from_numpy(np.ascontiguousarray(vol[list(slices)])).to(dev)
    n = len(slices)
    assert ops.mask_regions_fits(512, 512, 64)
    new = ct.phantom_roi_batch(x, 0.5)
    reg = ct.get_regions_batch(x, 0.5, True, True, 64)
    disk = ct._disk_on_device(512, 512, 0.5, x.device)
    spans = ct._disk_spans_on_device(512, 512, 0.5, x.device)
    assert np.array_equal(ct.row_spans(disk.cpu().numpy()), spans.cpu().numpy())
    p32, rawmax, lo, hi = ops.edge_plane(x, 1, spans=spans)
    assert torch.equal(p32, reg["edges"].to(torch.float32))
    lo2, hi2 = ops.minmax_masked(reg["edges"], disk)
    assert torch.equal(lo, lo2) and torch.equal(hi, hi2) and torch.equal(rawmax, ops.minmax(ops.scharr(x))[1])
    thr, otsu = ops.edge_otsu(p32, lo, hi, frames=x, sigma=1, spans=spans, scale=0.8)
    assert torch.equal(otsu, reg["otsu"])
    catphan_size = np.pi * 101**2 / 0.5**2
    r = ops.edge_regions(p32, x, 1, thr, 4, True, 64, catphan_size=catphan_size, rawmax=rawmax, return_mask=True)
    assert not r["status"].any() and torch.equal(r["count"], reg["num"]) and torch.equal(r["mask"], reg["bw"])
    for i in range(n):
        k = int(r["count"][i])
        assert torch.equal(r["table"][i, :k], reg["stats"][i, :k, :7]), i
    want = ct._select_phantom_roi(reg["stats"].cpu().numpy(), np.minimum(reg["num"].cpu().numpy(), 64),
                                  reg["overflow"].cpu().numpy(), rawmax.cpu().numpy(), catphan_size, 64)
    assert np.array_equal(r["roi"].cpu().numpy(), want, equal_nan=True) and np.array_equal(new, want, equal_nan=True)
    assert (new[:, 0] == 0).all() and np.all(np.abs(new[:, 3:5] - 255.5) < 8)
    # the other status codes of the on-device choice: wrong size (3), no edges (1), label table too small (4), no region (2)
    for size, rmx, ml, code in ((catphan_size * 2, rawmax, 64, 3), (catphan_size, torch.zeros_like(rawmax), 64, 1),
                                (catphan_size, rawmax, 1, None)):
        q = ops.edge_regions(p32, x, 1, thr, 4, True, ml, catphan_size=size, rawmax=rmx)
        w2 = ct._select_phantom_roi(q["table"].cpu().numpy(), np.minimum(q["count"].cpu().numpy(), ml),
                                    (q["count"].cpu().numpy() > ml).astype(np.int32), rmx.cpu().numpy(), size, ml)
        assert np.array_equal(q["roi"].cpu().numpy(), w2, equal_nan=True)
        if code is not None:
            assert (w2[:, 0] == code).all()
    q = ops.edge_regions(p32, x, 1, torch.full_like(hi, float("inf")), 4, True, 64, catphan_size=catphan_size, rawmax=rawmax)
    assert (q["roi"][:, 0] == 2).all()
    # thresholds placed ON plane values: the float32 bracket of those pixels straddles the threshold, so they are decided by
    # the exact recomputation; the mask must equal the float64 comparison
    e64 = reg["edges"]
    tt = e64.reshape(n, -1).sort(dim=1).values[:, -500].contiguous()     # high: a mask of few runs
    q = ops.edge_regions(p32, x, 1, tt, 0, False, 64, return_mask=True, want_table=False)
    assert not q["status"].any() and torch.equal(q["mask"], (e64 > tt[:, None, None]).to(torch.uint8))
    # round 6: the comparison runs in the float32 BIT domain (bits(v) against bits(a), a = the largest float32 <= t): every
    # position of t relative to the float32 grid, the ends of the range, and thresholds nothing can pass
    f32 = tt.to(torch.float32)
    up = torch.nextafter(f32, torch.full_like(f32, float("inf"))).to(torch.float64)
    dn = torch.nextafter(f32, torch.full_like(f32, float("-inf"))).to(torch.float64)
    variants = [f32.to(torch.float64), up, dn, (up + f32.to(torch.float64)) / 2, torch.nextafter(tt, torch.full_like(tt, float("inf"))),
                torch.nextafter(tt, torch.full_like(tt, float("-inf"))), torch.zeros_like(tt), torch.full_like(tt, -1.0),
                torch.full_like(tt, 1e-300), torch.full_like(tt, 1e-45), torch.full_like(tt, 3.5e38), torch.full_like(tt, 1e300),
                torch.full_like(tt, float("nan")), e64.reshape(n, -1).min(dim=1).values, e64.reshape(n, -1).max(dim=1).values]
    for k, tv in enumerate(variants):
        q = ops.edge_regions(p32, x, 1, tv.contiguous(), 0, False, 64, return_mask=True, want_table=False)
        assert torch.equal(q["mask"], (e64 > tv[:, None, None]).to(torch.uint8)), k
    return new


def check_histogram16_one_read(dev, sizes=((512, 512), (513, 520), (600, 437), (505, 523))):
    """pl_hist16 on frames of at least 2^18 pixels (the single-read two-window kernel) against np.bincount: bimodal frames
    whose range exceeds the two LDS windows (pixels in between take the global-atomic path), a clipped noisy background (the
    hot-value peel), narrow ranges (one contiguous window), constants, int16, a size that is not a multiple of 8."""
    import torch

    from pylinac_amd import ops

    rng = np.random.default_rng(61)
    n_checked = 0
    for h, w in sizes:
        yy, xx = np.mgrid[:h, :w]
        field = ((abs(yy - h / 2) < h / 5) & (abs(xx - w / 2) < w / 4)).astype(np.float64)
        from scipy import ndimage
        soft = ndimage.gaussian_filter(field, 6)
        frames = [
            np.clip(soft * 60000 + rng.normal(0, 65, (h, w)), 0, 65535),                 # clipped dark noise + field + penumbra
            np.clip(soft * 52000 + 9000 + rng.normal(0, 40, (h, w)), 0, 65535),          # offset background: no single hot value
            np.clip(20000 + rng.normal(0, 300, (h, w)), 0, 65535),                       # narrow range: one contiguous window
            rng.integers(0, 65536, (h, w)).astype(np.float64),                            # everything everywhere
            np.full((h, w), 777.0), np.zeros((h, w)), np.full((h, w), 65535.0),
            np.where(rng.random((h, w)) < 0.5, 3.0, 65000.0),                            # two values
        ]
        a = np.stack(frames).astype(np.uint16)
        a[1, 0, 0] = 0                       # extremes the 1/16 row sample does not see
        a[1, h - 1, w - 1] = 65535
        for arr in (a, (a.astype(np.int32) - 32768).astype(np.int16)):
            got = ops.histogram16(torch.from_numpy(arr).to(dev)).cpu().numpy().view(np.uint32)
            for i in range(arr.shape[0]):
                want = np.bincount(arr[i].astype(np.int64).ravel() + (32768 if arr.dtype == np.int16 else 0), minlength=65536)
                assert np.array_equal(got[i], want), (h, w, arr.dtype, i, np.flatnonzero(got[i] != want)[:5])
                n_checked += 1
    return n_checked


def check_hist16_wl_many_frames(dev, n=260, h=512, w=512):
    """pl_hist16_wl on MORE frames than the chip has CUs (the launcher then takes the two-workgroups-per-CU instantiation with
    9 728-bin windows) against the kernels that do the same work apart: pl_hist16 (19 456-bin windows), pl_order_stats,
    pl_edge_minmax, and the true tile maxima -- Winston-Lutz-like frames (flat background, field, penumbra, clipped noise),
    frames with every value of the range (most pixels BETWEEN the small windows: the global-atomic path), int16."""
    import torch
    from scipy import ndimage

    from pylinac_amd import ops

    rng = np.random.default_rng(83)
    yy, xx = np.mgrid[:h, :w]
    field = ndimage.gaussian_filter(((abs(yy - h / 2) < h / 6) & (abs(xx - w / 2) < w / 6)).astype(np.float64), 4)
    kinds = [np.clip(field * 52000 + 3000, 0, 65535),                                      # noise-free
             np.clip(field * 52000 + rng.normal(0, 65, (h, w)), 0, 65535),                  # clipped dark noise
             rng.integers(0, 65536, (h, w)).astype(np.float64),                             # everything everywhere
             np.clip(30000 + rng.normal(0, 4000, (h, w)), 0, 65535)]                        # one broad mode
    base = np.stack(kinds).astype(np.uint16)
    arr = np.empty((n, h, w), dtype=np.uint16)
    for i in range(n):
        arr[i] = np.roll(base[i % 4], (i, 3 * i), axis=(0, 1))
    arr[1, 0, 0], arr[1, -1, -1] = 0, 65535
    cnt = h * w
    ranks = np.array([0, cnt - 1, cnt // 2, int(0.999 * cnt), 12345, 7 * cnt // 10, int(0.05 * cnt), 1, cnt - 2, int(0.3 * cnt),
                      int(0.5001 * cnt), int(0.9 * cnt)], dtype=np.int64)
    for a in (arr, (arr.astype(np.int32) - 32768).astype(np.int16)):
        x = torch.from_numpy(a).to(dev)
        _, tmax, emin, emax, st = ops.histogram16(x, tiles=True, edge_window=2, ranks=ranks)   # (the table is scratch in this form)
        assert torch.equal(st, ops.order_stats(x, ranks, hist=ops.histogram16(x))), "order statistics"
        srt = np.sort(a[:8].reshape(8, -1).astype(np.int64), axis=1) if n >= 8 else np.sort(a.reshape(n, -1).astype(np.int64), axis=1)
        assert np.array_equal(st[:len(srt)].cpu().numpy(), srt[:, ranks]), "order statistics vs numpy"
        e0, e1 = ops.edge_minmax(x, 2)
        assert torch.equal(emin, e0) and torch.equal(emax, e1), "edge strips"
        keys = a.astype(np.int64) + (32768 if a.dtype == np.int16 else 0)
        true_max = keys.reshape(n, -1, 512).max(axis=2)
        got = tmax.cpu().numpy().astype(np.int64) & 0xFFFF
        assert ((got == true_max) | (got == 0xFFFF)).all() and (got == true_max).mean() > 0.99
    return 2 * n


def check_fused_tail_vs_separate(dev, shapes=((3, 200, 520), (2, 128, 64), (2, 130, 1032), (1, 2, 8))):
    """pl_median3_threshold_colparts_u16 + pl_colparts_profile_fwxm (the EPID pipeline's third stage and its one-launch tail)
    == pl_median3_threshold_colsum_u16 -> pl_colsum_to_mean -> pl_find_peaks -> pl_fwxm_record, bit for bit, on heights that
    leave a partial band and widths that leave a partial column group."""
    import ctypes as C

    import torch

    from pylinac_amd import _lib, ops
    from pylinac_amd._lib import check

    lib = _lib.load()
    rng = np.random.default_rng(17)
    st = torch.cuda.current_stream().cuda_stream
    for n, h, w in shapes:
        yy, xx = np.mgrid[:h, :w]
        base = 20000 * np.exp(-0.5 * ((xx - w * 0.55) / (w * 0.18)) ** 2) + 3000
        fr = np.clip(base[None] + rng.normal(0, 400, (n, h, w)), 0, 65535).astype(np.uint16)
        x = torch.from_numpy(fr).to(dev)
        thr = torch.from_numpy(np.array([9000, 0, 70000][:n] if n <= 3 else [9000] * n, dtype=np.int32)).to(dev)
        prm = ops.make_peak_params(w, fwxm_height=0.5, max_number=1)

        def outputs():
            return dict(out=torch.empty_like(x), prof=torch.empty((n, w), dtype=torch.float64, device=dev),
                        cnt=torch.empty(n, dtype=torch.int32, device=dev), idx=torch.empty((n, 1), dtype=torch.int32, device=dev),
                        lb=torch.empty((n, 1), dtype=torch.int32, device=dev), rb=torch.empty((n, 1), dtype=torch.int32, device=dev),
                        props=torch.empty((n, 6, 1), dtype=torch.float64, device=dev),
                        status=torch.empty(n, dtype=torch.int32, device=dev), fwxm=torch.empty((n, 8), dtype=torch.float64, device=dev))

        a, b = outputs(), outputs()
        colsum = torch.empty((n, w), dtype=torch.int64, device=dev)
        check(lib.pl_median3_threshold_colsum_u16(x.data_ptr(), a["out"].data_ptr(), n, h, w, thr.data_ptr(), colsum.data_ptr(), st), "colsum")
        check(lib.pl_colsum_to_mean(colsum.data_ptr(), n, w, h, a["prof"].data_ptr(), st), "mean")
        check(lib.pl_find_peaks(a["prof"].data_ptr(), n, w, w, C.byref(prm), 1, a["cnt"].data_ptr(), a["idx"].data_ptr(), a["lb"].data_ptr(),
                                a["rb"].data_ptr(), a["props"].data_ptr(), a["status"].data_ptr(), st), "peaks")
        check(lib.pl_fwxm_record(a["cnt"].data_ptr(), a["idx"].data_ptr(), a["props"].data_ptr(), 1, n, a["fwxm"].data_ptr(), st), "fwxm")
        bands = -(-h // lib.pl_colparts_band_rows())
        parts = torch.full((n, bands, w), -1, dtype=torch.int32, device=dev)       # every element must be overwritten
        check(lib.pl_median3_threshold_colparts_u16(x.data_ptr(), b["out"].data_ptr(), n, h, w, thr.data_ptr(), parts.data_ptr(), st), "colparts")
        check(lib.pl_colparts_profile_fwxm(parts.data_ptr(), n, bands, w, h, C.byref(prm), 1, b["prof"].data_ptr(), b["cnt"].data_ptr(),
                                           b["idx"].data_ptr(), b["lb"].data_ptr(), b["rb"].data_ptr(), b["props"].data_ptr(),
                                           b["status"].data_ptr(), b["fwxm"].data_ptr(), st), "tail")
        assert np.array_equal(parts.cpu().numpy().view(np.uint32).astype(np.int64).sum(1), colsum.cpu().numpy()), (n, h, w)
        for k in ("out", "prof", "cnt", "status", "fwxm"):
            assert np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), equal_nan=k == "fwxm"), (n, h, w, k)
        c = a["cnt"].cpu().numpy()
        for k in ("idx", "lb", "rb"):
            assert np.array_equal(a[k].cpu().numpy()[c > 0], b[k].cpu().numpy()[c > 0]), (n, h, w, k)
        assert np.array_equal(a["props"].cpu().numpy()[c > 0], b["props"].cpu().numpy()[c > 0]), (n, h, w)
    return len(shapes)


def _gamma_geometric_cases(g):
    import json

    for k in range(int(g["count"])):
        kw = json.loads(str(g[f"kw{k}"]))
        for name in ("reference", "evaluation", "reference_coordinates", "evaluation_coordinates"):
            if f"{name}{k}" in g.files:
                kw[name] = g[f"{name}{k}"]
        yield k, kw, g[f"gamma{k}"]


def check_gamma_geometric(golden, dev):
    """pl_gamma_geometric against the reference's own gamma_geometric (its known-answer inputs from
    tests_basic/core/test_gamma.py:304-372; dose-like profile pairs incl. coarser, reversed and integer evaluation samples) and
    PhysicalProfileMixin.gamma against the reference's own method (tests/golden/gamma_geometric.npz): NaN / fill positions
    identical, gamma to 1e-12 (float64 in the reference's operation order; only BLAS ``dot`` / ``math.dist`` may round the last
    bit differently); the reference's ValueErrors."""
    from pylinac_amd import gamma as pg
    from pylinac_amd import profile as pp

    g = golden("gamma_geometric")
    for k, kw, want in _gamma_geometric_cases(g):
        got = pg.gamma_geometric(device=dev, **kw)
        assert np.array_equal(np.isnan(got), np.isnan(want)), k
        assert np.allclose(got, want, rtol=1e-12, atol=1e-13, equal_nan=True), (k, np.nanmax(np.abs(got - want)))
    for bad in (dict(reference=np.ones((2, 2)), evaluation=np.ones(4)), dict(reference=np.ones(5), evaluation=np.ones(5), distance_to_agreement=0),
                dict(reference=np.ones(5), evaluation=np.ones(5), dose_to_agreement=-1),
                dict(reference=np.ones(5), evaluation=np.ones(5), reference_coordinates=np.arange(6)),
                dict(reference=np.ones(5), evaluation=np.ones(5), evaluation_coordinates=np.array([0.0, 1, 3, 2, 4]))):
        with pytest.raises(ValueError):
            pg.gamma_geometric(device=dev, **bad)
    # the profile method: both profiles re-centred on their geometric centres, physical x-values
    r1, e1 = pp.FWXMProfilePhysical(g["p.ref"], dpmm=2.0), pp.FWXMProfilePhysical(g["p.ev"], dpmm=2.0)
    g1, rr, ee = r1.gamma(e1, dose_to_agreement=1, distance_to_agreement=1, return_profiles=True)
    assert np.allclose(g1, g["p.gamma1"], rtol=1e-12, atol=1e-13, equal_nan=True)
    assert np.array_equal(rr.x_values, g["p.ref_x1"]) and np.array_equal(ee.x_values, g["p.ev_x1"])
    assert np.array_equal(r1.x_values, np.arange(len(g["p.ref"])))                       # the originals are untouched
    r2 = pp.FWXMProfilePhysical(g["p.ref"], dpmm=2.0)
    e2 = pp.FWXMProfilePhysical(g["p.ev2"], x_values=g["p.xs2"] + 0.3, dpmm=None)
    g2 = r2.gamma(e2, dose_to_agreement=2, distance_to_agreement=2, gamma_cap_value=3, dose_threshold=8, fill_value=-1.0)
    assert np.allclose(g2, g["p.gamma2"], rtol=1e-12, atol=1e-13)
    with pytest.raises(ValueError, match="must also be a physical profile"):
        r1.gamma(pp.FWXMProfile(g["p.ev"]))


def _pf_mlc_cases(g):
    for name, mlc, tr in zip(g["names"], g["mlcs"], g["transposed"]):
        yield str(name), str(mlc), ("LEFT_RIGHT" if bool(tr) else "UP_DOWN")


def check_pf_mlc_oracle(g):
    """oracle.pf_measure(mlc=) against the reference's OWN PicketFence.analyze(mlc=MLC.X) on the HD Millennium, Agility,
    Halcyon-distal and B-mod banks (tests/golden/make_pf_mlc_golden.py): spacing, picket indices, and every position the
    reference kept, bit for bit."""
    from oracle import pylinac_oracle as o

    for name, mlc, orient in _pf_mlc_cases(g):
        raw, dpmm = g[f"{name}.cropped"], float(g[f"{name}.dpmm"])
        r = o.pf_measure(o.normalize(o.ground(raw)), dpmm, mlc=mlc, orientation=orient)
        assert r["spacing"] == float(g[f"{name}.spacing"]), name
        idx = {n: i for i, (n, c, w) in enumerate(r["leaves"])}
        meas = g[f"{name}.meas"]
        assert len(meas) > 150
        for row in meas:
            leaf, picket = int(row[0]), int(row[1])
            assert r["peak_idxs"][picket] == row[2] and r["position"][idx[leaf], picket] == row[3], (name, leaf, picket)


def check_pf_mlc_device(g, dev):
    """picketfence.analyze_batch(mlc=) == the oracle on every window (NaN pattern included) and == the reference's own
    analyze() on every window it kept, for the leaf banks other than the Millennium 120."""
    from oracle import pylinac_oracle as o
    from pylinac_amd import picketfence as ppf

    rejected = 0
    for name, mlc, orient in _pf_mlc_cases(g):
        raw, dpmm = g[f"{name}.cropped"], float(g[f"{name}.dpmm"])
        res = ppf.analyze_batch(torch.from_numpy(np.ascontiguousarray(raw)[None]).to(dev), dpmm, mlc=mlc, orientation=orient)
        ref = o.pf_measure(o.normalize(o.ground(raw)), dpmm, mlc=mlc, orientation=orient)
        P = len(ref["peak_idxs"])
        assert int(res.picket_count[0]) == P and float(res.spacing[0]) == ref["spacing"] == float(g[f"{name}.spacing"]), name
        assert np.array_equal(res.picket_idx[0, :P].cpu().numpy(), ref["peak_idxs"]) and res.leaf_nums == [n for n, _, _ in ref["leaves"]]
        got = res.position[0, :, :P].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref["position"])), name
        assert np.array_equal(got[~np.isnan(got)], ref["position"][~np.isnan(got)]), name
        assert not (res.status[0, :, :P] == 3).any(), name                 # no window was refused as too tall
        # the edge test decided from integer row moments (default) == numpy's float64 deviations evaluated for every window
        ex = ppf.analyze_batch(torch.from_numpy(np.ascontiguousarray(raw)[None]).to(dev), dpmm, mlc=mlc, orientation=orient,
                               exact_deviation=True)
        assert torch.equal(ex.status, res.status) and torch.equal(torch.nan_to_num(ex.position, nan=-1.0), torch.nan_to_num(res.position, nan=-1.0)), name
        rejected += int((res.status[0, :, :P] == 2).sum())
        idx = {n: i for i, n in enumerate(res.leaf_nums)}
        for row in g[f"{name}.meas"]:
            assert float(res.position[0, idx[int(row[0])], int(row[1])]) == row[3], (name, row[:2])
    assert rejected > 0                                                     # (jaw-blocked rows: windows that fail the test)
    # a window ON the decision boundary: edge_threshold := numpy's own max(std) / median(std) of one window, so that the integer
    # moments cannot decide it (inside the 1e-7 margin) and numpy's float64 sequence must -- the oracle says how it comes out
    name, mlc, orient = next(iter(_pf_mlc_cases(g)))
    raw, dpmm = g[f"{name}.cropped"], float(g[f"{name}.dpmm"])
    img = o.normalize(o.ground(raw))
    base = o.pf_measure(img, dpmm, mlc=mlc, orientation=orient)
    ratios = []
    sp, peaks = base["spacing"], base["peak_idxs"]
    for num, center, width in base["leaves"]:                 # _get_mlc_window (picketfence.py:859-886), UP_DOWN
        c_px, w_px = center * dpmm + raw.shape[0] / 2, width * dpmm
        lo, hi = max(int(c_px - w_px / 2), 0), min(int(c_px + w_px / 2), raw.shape[0])
        for pk in peaks[:3]:
            t0, t1 = max(int(pk - sp / 2), 0), min(int(pk + sp / 2), raw.shape[1])
            std = np.std(img[lo:hi, t0:t1], axis=1)
            if len(std) and np.median(std) > 0:
                ratios.append(max(std) / np.median(std))
    for thr in sorted(ratios)[len(ratios) // 2 - 1:len(ratios) // 2 + 2]:
        want = o.pf_measure(img, dpmm, mlc=mlc, orientation=orient, edge_threshold=float(thr))
        got = ppf.analyze_batch(torch.from_numpy(np.ascontiguousarray(raw)[None]).to(dev), dpmm, mlc=mlc, orientation=orient,
                                edge_threshold=float(thr)).position[0, :, :len(peaks)].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want["position"])), thr
        assert 0 < int(np.isnan(want["position"]).sum()) < want["position"].size       # the threshold splits the windows


def check_pf_other_dtypes(g, dev):
    """picketfence.analyze_batch on int16 frames and on float64 frames holding integers -- what the
    reference's loader produces for a signed panel, for ``dtype=float`` and for integer rescale tags -- against the oracle run
    on exactly those arrays (``normalize(ground(a))``: positions bit for bit, same NaN pattern) and against the uint16 result
    of the same pixels; frames that cannot be bridged exactly (non-integer values, an int16 range beyond 32767 where the
    reference's own ground() wraps) come back refused: status 3, NaN positions."""
    from oracle import pylinac_oracle as o
    from pylinac_amd import picketfence as ppf

    name, mlc, orient = next(iter(_pf_mlc_cases(g)))
    raw, dpmm = np.ascontiguousarray(g[f"{name}.cropped"]), float(g[f"{name}.dpmm"])
    base = ppf.analyze_batch(torch.from_numpy(raw[None]).to(dev), dpmm, mlc=mlc, orientation=orient)
    half = (raw.astype(np.int32) // 2).astype(np.uint16)                    # fits int16 after the shift below
    variants = {
        "float64": raw.astype(np.float64),
        "float64 + intercept": raw.astype(np.float64) - 1024.0,
        "int16": (half.astype(np.int32) - 20000).astype(np.int16),
    }
    for tag, arr in variants.items():
        res = ppf.analyze_batch(torch.from_numpy(arr[None]).to(dev), dpmm, mlc=mlc, orientation=orient)
        ref = o.pf_measure(o.normalize(o.ground(arr)), dpmm, mlc=mlc, orientation=orient)
        P = len(ref["peak_idxs"])
        got = res.position[0, :, :P].cpu().numpy()
        assert int(res.picket_count[0]) == P and float(res.spacing[0]) == ref["spacing"], tag
        assert np.array_equal(np.isnan(got), np.isnan(ref["position"])), tag
        assert np.array_equal(got[~np.isnan(got)], ref["position"][~np.isnan(got)]), tag
        if tag.startswith("float64"):
            assert torch.equal(torch.nan_to_num(res.position, nan=-1.0), torch.nan_to_num(base.position, nan=-1.0)), tag
            assert torch.equal(res.status, base.status)
    # refused frames inside a batch: the others are measured as usual
    frac = raw.astype(np.float64)
    frac[3, 5] += 0.5
    batch = np.stack([raw.astype(np.float64), frac])
    res = ppf.analyze_batch(torch.from_numpy(batch).to(dev), dpmm, mlc=mlc, orientation=orient)
    assert torch.equal(res.status[0], base.status[0]) and bool((res.status[1] == 3).all()) and bool(torch.isnan(res.position[1]).all())
    wide = np.where(raw > raw.mean(), 30000, -30000).astype(np.int16)
    res = ppf.analyze_batch(torch.from_numpy(wide[None]).to(dev), dpmm, mlc=mlc, orientation=orient)
    assert bool((res.status == 3).all())
    with pytest.raises(TypeError):
        ppf.analyze_batch(torch.from_numpy(raw.astype(np.float32)[None]).to(dev), dpmm, mlc=mlc, orientation=orient)


def check_fwxm_short_profiles(dev, trials=600):
    """The FWXM search (max_number = 1 by prominence) on short one-wave profiles (the picket-fence windows' case:
    peaks_device.h) against scipy through the oracle, on profiles built for its corner cases: smooth humps, the global
    maximum on an edge sample or shared by two peaks, plateaus, a side peak more prominent than the highest one, monotone and
    constant profiles.  Indices, bases, prominence, width, width height and both interpolated positions bit for bit."""
    from oracle import pylinac_oracle as o
    from pylinac_amd import ops
    from scipy.signal import find_peaks as sp_find_peaks

    rng = np.random.default_rng(31)
    found = skipped = 0
    for t in range(trials):
        L = int(rng.integers(3, 65))
        kind = t % 6
        xx = np.arange(L, dtype=float)
        if kind == 0:                                         # one hump + noise (a picket-fence window)
            x = np.exp(-0.5 * ((xx - rng.uniform(0.2, 0.8) * L) / rng.uniform(1.5, 6)) ** 2) + rng.normal(0, 0.01, L)
        elif kind == 1:                                       # integers: plateaus, equal maxima
            x = rng.integers(0, 5, L).astype(float)
        elif kind == 2:                                       # the highest sample sits on an edge
            x = np.abs(rng.normal(size=L)).cumsum() * (1 if rng.random() < 0.5 else -1)
            x = x - x.min()
            x[rng.integers(1, L - 1) if L > 2 else 0] += 0.3
        elif kind == 3:                                       # a low side peak next to a deep valley beats the high peak on a shoulder
            x = np.concatenate([[0.0, 5.0, 0.0], np.linspace(5.5, 6.0, max(L - 5, 1)), [5.9, 5.95]])[:L] + rng.normal(0, 1e-3, min(L, max(L - 5, 1) + 5))
        elif kind == 4:
            x = rng.random(L)
        else:
            x = np.full(L, 2.0) if t % 12 == 5 else np.sort(rng.random(L))
        x = np.ascontiguousarray(x[:L] if len(x) >= L else np.pad(x, (0, L - len(x))))
        h = float(rng.choice([0.5, 0.2, 0.8]))
        pk, props = sp_find_peaks(x, prominence=0)
        if len(pk) > 1:
            pr = np.sort(props["prominences"])
            if pr[-1] == pr[-2]:                              # an exact tie for the first place: platform-defined in the reference
                skipped += 1
                continue
        i1, p1 = o.find_peaks(x, fwxm_height=h, max_number=1)
        i2, p2 = ops.find_peaks_batch(torch.from_numpy(x).to(dev), fwxm_height=h, max_number=1).to_host(0)
        assert np.array_equal(i1, i2), (t, kind, x.tolist())
        for k in p1:
            assert np.array_equal(p1[k], p2[k]), (t, kind, k, x.tolist())
        found += len(i1)
    assert found > 0.6 * (trials - skipped) and skipped < trials // 4, (found, skipped)


def check_field_cax_tile_maxima(dev, big=False):
    """pl_hist16_tiles + pl_field_cax_tiles against pl_hist16 + pl_field_cax: identical histograms, tile maxima that are the
    true maxima of their 512-pixel tiles (or 0xffff), identical field CAX records -- uint16 and int16 frames, thresholds below
    every pixel / above every pixel / inside the noise, a frame with holes, and frames the tiled pass must refuse (pixel count
    not a multiple of 512: the full pass runs, same results)."""
    from pylinac_amd import ops

    rng = np.random.default_rng(77)
    h, w = (1024, 1024) if big else (512, 512)              # >= 2^18 pixels: the two-window histogram kernel with tile maxima
    base = rng.integers(900, 1100, (5, h, w)).astype(np.int64)
    base[4] = rng.integers(0, 65536, (h, w))                # every value of the range: most order statistics sit BETWEEN the
    #                                                         histogram kernel's two LDS windows (the table's own bins)
    base[0, 200:260, 300:380] += 30000                      # a field
    base[1, 100:400, 50:450] += 20000
    base[1, 200:230, 200:260] -= 20000                      # ... with a hole
    base[2, 5:9, 7:11] += 40000                             # a tiny one, touching nothing
    frames = np.clip(base, 0, 65535).astype(np.uint16)
    done = 0
    for dtype, arr in ((torch.uint16, frames), (torch.int16, (frames.astype(np.int64) - 20000).astype(np.int16))):
        x = torch.from_numpy(arr).to(dev)
        hist0 = ops.histogram16(x)
        hist1, tmax = ops.histogram16(x, tiles=True)
        assert torch.equal(hist0, hist1)
        for win in ((1, 2, 5) if big else (2,)):               # pl_hist16_wl: the edge strips' extrema from the same launch
            h2, t2, emin, emax = ops.histogram16(x, tiles=True, edge_window=win)
            e0, e1 = ops.edge_minmax(x, win)
            assert torch.equal(h2, hist0) and torch.equal(t2, tmax) and torch.equal(emin, e0) and torch.equal(emax, e1), (str(dtype), win)
            a = arr.astype(np.int64)
            strips = [np.concatenate([a[k, :win].ravel(), a[k, -win:].ravel(), a[k, :, :win].ravel(), a[k, :, -win:].ravel()]) for k in range(len(a))]
            assert emin.cpu().tolist() == [int(v.min()) for v in strips] and emax.cpu().tolist() == [int(v.max()) for v in strips]
        # ... and the order statistics selected inside the same launch == pl_order_stats_from_hist == numpy's sort
        cnt = arr[0].size
        ranks = np.array([0, cnt - 1, 1, cnt // 2, cnt // 2 + 1, int(0.999 * cnt), 12345, cnt - 2, 7 * cnt // 10, -5, cnt + 9], dtype=np.int64)
        _, t3, emin3, emax3, st = ops.histogram16(x, tiles=True, edge_window=2, ranks=ranks)
        want = ops.order_stats(x, ranks, hist=hist0)
        assert torch.equal(st, want) and torch.equal(t3, tmax)
        srt = np.sort(arr.reshape(len(arr), -1).astype(np.int64), axis=1)
        assert np.array_equal(st.cpu().numpy(), srt[:, np.clip(ranks, 0, cnt - 1)])
        keys = arr.astype(np.int64) + (32768 if dtype == torch.int16 else 0)
        true_max = keys.reshape(len(arr), -1, 512).max(axis=2)
        got = tmax.cpu().numpy().astype(np.int64) & 0xFFFF
        assert ((got == true_max) | (got == 0xFFFF)).all() and (got == true_max).mean() > 0.99
        vmin = torch.from_numpy(arr.reshape(len(arr), -1).min(axis=1).astype(np.float64)).to(dev)
        vmax = torch.from_numpy(arr.reshape(len(arr), -1).max(axis=1).astype(np.float64)).to(dev)
        for thr in (0.5, 0.0, 1.5, 0.002, 0.9999):            # mid-level, everything, nothing, inside the noise, only the peak
            t = torch.full((len(arr),), thr, dtype=torch.float64, device=dev)
            a, sa = ops.field_cax(x, vmin, vmax - vmin, t, defer=True)
            b, sb = ops.field_cax(x, vmin, vmax - vmin, t, defer=True, tile_max=tmax)
            ok = sa == 0                                     # (status 1 = window too large: the record is not written)
            assert torch.equal(sa, sb) and torch.equal(torch.nan_to_num(a[ok], nan=-1.0), torch.nan_to_num(b[ok], nan=-1.0)), (str(dtype), thr)
            done += int(ok.sum())
        if dtype == torch.uint16:                            # a divisor that is not positive keeps the float64 test on every tile
            t = torch.full((len(arr),), 0.5, dtype=torch.float64, device=dev)
            a, sa = ops.field_cax(x, vmin, -(vmax - vmin), t, defer=True)
            b, sb = ops.field_cax(x, vmin, -(vmax - vmin), t, defer=True, tile_max=tmax)
            ok = sa == 0
            assert torch.equal(sa, sb) and torch.equal(torch.nan_to_num(a[ok], nan=-1.0), torch.nan_to_num(b[ok], nan=-1.0))
    assert done >= 20
    # 513 x 520: two-window kernel (>= 2^18 pixels), but 266 760 pixels are not a whole number of tiles -> full pass, same record
    odd = np.clip(rng.integers(900, 1100, (2, 513, 520)), 0, 65535).astype(np.uint16)
    odd[:, 100:180, 90:200] += 30000
    x = torch.from_numpy(odd).to(dev)
    _, tmax = ops.histogram16(x, tiles=True)
    _, _, emin, emax = ops.histogram16(x, tiles=True, edge_window=2)
    e0, e1 = ops.edge_minmax(x, 2)
    assert torch.equal(emin, e0) and torch.equal(emax, e1)
    # rows of 100 vectors (not a power of two, the stream ends in tail vectors): the strips come out of the histogram's own
    # loads for windows up to a vector wide, out of its prologue beyond that -- int16 too
    for dt in ((np.uint16, np.int16) if big else (np.int16,)):            # (the CPU emulator: one dtype, three windows)
        wide = rng.integers(0, 65536, (2, 330, 800)).astype(np.uint16).view(dt) if dt == np.int16 else rng.integers(0, 65536, (2, 330, 800)).astype(dt)
        wide[0, 3:-3, 3:-3] = np.clip(wide[0, 3:-3, 3:-3].astype(np.int64), -20000, 20000).astype(dt)   # extrema only in the strips
        xw = torch.from_numpy(wide).to(dev)
        for win in ((1, 2, 3, 8, 9) if big else (2, 8, 9)):
            hw, tw, emin, emax = ops.histogram16(xw, tiles=True, edge_window=win)
            a = wide.astype(np.int64)
            strips = [np.concatenate([a[k, :win].ravel(), a[k, -win:].ravel(), a[k, :, :win].ravel(), a[k, :, -win:].ravel()]) for k in range(2)]
            assert emin.cpu().tolist() == [int(v.min()) for v in strips] and emax.cpu().tolist() == [int(v.max()) for v in strips], (dt.__name__, win)
            assert not big or torch.equal(hw, ops.histogram16(xw))
    small = torch.from_numpy(odd[:, :100, :120].copy()).to(dev)            # below 2^18 pixels: the multi-part histogram + the stand-alone edge kernel
    _, ts, emin, emax = ops.histogram16(small, tiles=True, edge_window=2)
    e0, e1 = ops.edge_minmax(small, 2)
    assert torch.equal(emin, e0) and torch.equal(emax, e1) and bool(((ts.to(torch.int32) & 0xFFFF) == 0xFFFF).all())
    rk = np.array([0, 11999, 6000, 3], dtype=np.int64)
    st = ops.histogram16(small, tiles=True, edge_window=2, ranks=rk)[4]    # ... + the stand-alone order-statistics kernel
    assert np.array_equal(st.cpu().numpy(), np.sort(odd[:, :100, :120].reshape(2, -1).astype(np.int64), axis=1)[:, rk])
    vmin = torch.from_numpy(odd.reshape(2, -1).min(axis=1).astype(np.float64)).to(dev)
    rng_ = torch.from_numpy((odd.reshape(2, -1).max(axis=1) - odd.reshape(2, -1).min(axis=1)).astype(np.float64)).to(dev)
    t = torch.full((2,), 0.5, dtype=torch.float64, device=dev)
    assert torch.equal(ops.field_cax(x, vmin, rng_, t), ops.field_cax(x, vmin, rng_, t, tile_max=tmax))


def check_bb_sweep_run_table_tiers(dev, full=True):
    """pl_features_sweep's passes: windows whose speckle needs more row runs at a level than the first pass's table holds (896
    at this window size: three workgroups per CU), more than the second's (1 536), more than the third's (4 096: status 5, the
    caller's level-by-level path) -- every one gives the features of the level-by-level path, and the sweep alone
    (``defer=True``) reports status 0 for the first three."""
    from pylinac_amd import features as pf

    rng = np.random.default_rng(5)
    dpmm, n = 2.98, 140
    yy, xx = np.mgrid[0:n, 0:n].astype(float)
    wins = []
    for frac in (0.0, 0.06, 0.2, 0.5):
        img = np.full((n, n), 0.2) + rng.normal(0, 1e-4, (n, n))
        cy, cx = 70.3, 66.8
        far = np.hypot(yy - cy, xx - cx) > 16
        img[far & (rng.random((n, n)) < frac)] = 0.6
        img[np.hypot(yy - cy, xx - cx) < 2.5 * dpmm] = 1.0
        wins.append(img)
        runs = int(((img[:, 1:] > 0.3) & ~(img[:, :-1] > 0.3)).sum() + (img[:, 0] > 0.3).sum())
        assert {0.0: runs < 896, 0.06: 896 < runs <= 1536, 0.2: 1536 < runs <= 4096, 0.5: runs > 4096}[frac], (frac, runs)
    x = torch.from_numpy(np.stack(wins)).to(dev)
    alone = pf.find_features_batch(x, dpmm, 2.5, 0.5, defer=True)
    assert alone["status"].cpu().tolist() == [0, 0, 0, 5]
    assert alone["count"].cpu().tolist()[:3] == [1, 1, 1]
    if not full:                                             # (the CPU emulator: the two windows that were handed on, once)
        lv = pf.find_features_batch(x[1:2], dpmm, 2.5, 0.5, level_by_level=True)
        assert torch.equal(alone["level"][1:2], lv["level"]) and torch.equal(alone["xy"][1:2, 0], lv["xy"][:, 0])
        assert torch.equal(alone["level"][2:3], lv["level"]) and bool((alone["xy"][2, 0] - lv["xy"][0, 0]).abs().max() < 0.05)   # same BB, other speckle
        return 2
    res = pf.find_features_batch(x, dpmm, 2.5, 0.5)
    lv = pf.find_features_batch(x, dpmm, 2.5, 0.5, level_by_level=True)
    assert res["count"].cpu().tolist() == [1, 1, 1, 1] == lv["count"].cpu().tolist()
    assert torch.equal(res["level"], lv["level"]) and torch.equal(res["xy"][:, 0], lv["xy"][:, 0])
    assert torch.equal(alone["xy"][:3, 0], lv["xy"][:3, 0])
    return 4


# ---- f1, the DICOM half ------------------------------------------------------------------------------------------------
def _dicom_cases(golden):
    g = golden("dicom")
    return {k[len("file__"):]: (g[k], g["expect__" + k[len("file__"):]]) for k in g.files if k.startswith("file__")}


def check_dicom_golden(golden, dev):
    """pylinac_amd.dicom (Part-10 walk on the host, pl_dicom_decode on the device) against the fixtures of
    tests/golden/make_dicom_golden.py: `pixel_array` of every case bit for bit (container dtype, pydicom 2.x semantics), the
    pydicom >= 3 unused-bit correction where BitsStored < BitsAllocated, `astype`, the rescaled / inverted float64 array of
    `DicomImage.__init__` (pylinac/core/image.py:1431-1444) against the oracle's restatement, and the metadata-driven
    properties (image.py:1491-1578)."""
    from oracle import pylinac_oracle as orc
    from pylinac_amd import dicom

    cases = _dicom_cases(golden)
    assert len(cases) >= 16
    for name, (blob, expect) in cases.items():
        meta, data = dicom.read_part10(blob.tobytes())
        want, tags, start = orc.dicom_pixel_array(blob)
        assert np.array_equal(want, expect) and want.dtype == expect.dtype, name      # the oracle against the encoded array
        assert meta.PixelData[0] == start, name
        frames, _ = dicom.load_frames([blob.tobytes()], raw_pixels=True, device=dev)
        got = dicom._to_numpy(frames)
        assert got.dtype == expect.dtype and np.array_equal(got.reshape(expect.shape), expect), name
        # DicomImage: rescale + inversion as the tags say
        img = dicom.DicomImage(blob.tobytes())
        ref = orc.dicom_image_array(blob)
        assert img.array.dtype == ref.dtype and np.array_equal(img.array, ref), name
        for dt in (np.float32, np.float64, np.int32, np.uint8):
            a = dicom.DicomImage(blob.tobytes(), dtype=dt, raw_pixels=True).array
            r = orc.dicom_image_array(blob, dtype=dt, raw_pixels=True)
            assert a.dtype == r.dtype and np.array_equal(a, r), (name, dt)
        for inv in (True, False):
            if inv and expect.dtype in (np.int8, np.uint32):
                with pytest.raises(NotImplementedError):
                    dicom.DicomImage(blob.tobytes(), invert_pixels=True)
                continue
            a = dicom.DicomImage(blob.tobytes(), invert_pixels=inv).array
            assert np.array_equal(a, orc.dicom_image_array(blob, invert_pixels=inv)), (name, inv)
        if "stored12" in name:
            fixed, _ = dicom.load_frames([blob.tobytes()], raw_pixels=True, correct_unused_bits=True, device=dev)
            w3, _, _ = orc.dicom_pixel_array(blob, correct_unused_bits=True)
            assert np.array_equal(dicom._to_numpy(fixed)[0], w3), name
            assert not np.array_equal(w3, expect)                                      # the fixture's unused bits are dirty
            lim = 4096 if expect.dtype == np.uint16 else 2048
            assert w3.max() < lim and w3.min() >= (0 if expect.dtype == np.uint16 else -2048)
    # the tags the image properties read
    img = dicom.DicomImage(cases["u16_epid_tags"][0].tobytes())
    assert img.sid == 1500.0 and img.sad == 1000.0
    assert abs(img.dpmm - (1 / 0.336) * 1.5) < 1e-12 and abs(img.dpi - img.dpmm * 25.4) < 1e-9
    cx, cy = img.center.x, img.center.y
    assert abs(img.cax.x - (cx - 1.5 * img.dpmm / 1.5)) < 1e-12 and abs(img.cax.y - (cy + -2.25 * img.dpmm / 1.5)) < 1e-12
    assert img.metadata.GantryAngle == 90.0 and img.metadata.get("RescaleSlope") is None
    plain = dicom.DicomImage(cases["u16_explicit"][0].tobytes(), dpi=100, sid=1200)
    assert plain.sid == 1200 and plain.dpi == 100 and abs(plain.dpmm - 100 / 25.4) < 1e-12 and plain.cax.x == plain.center.x
    # a CT-like stack: several files -> one batch, the fused rescale
    ct = cases["i16_ct"][0].tobytes()
    stack, metas = dicom.load_frames([ct, ct, ct], device=dev)
    assert stack.shape[0] == 3 and stack.dtype == torch.float64
    assert np.array_equal(stack[1].cpu().numpy(), orc.dicom_image_array(cases["i16_ct"][0]))
    # truncated Pixel Data: pydicom raises ValueError
    short = cases["u16_explicit"][0][:-200].tobytes()
    with pytest.raises((ValueError, struct_error())):
        dicom.load_frames([short], device=dev)


def struct_error():
    import struct

    return struct.error


def check_dicom_decode_fuzz(dev, frame_shapes=((9, 14), (16, 16)), n=3):
    """pl_dicom_decode on random byte buffers: every container width, both representations, both byte orders, each of the
    four alignments of the first frame, with / without the unused-bit correction, the three output forms -- against
    np.frombuffer (+ numpy's own shifts / casts), bit for bit."""
    from pylinac_amd import dicom

    rng = np.random.default_rng(77)
    for rows, cols in frame_shapes:
        for bits in (8, 16, 32):
            ib = bits // 8
            for rep in (0, 1):
                for big in (False, True):
                    for shift in range(4):
                        stored = int(rng.integers(max(2, bits - 7), bits + 1))
                        gap = int(rng.integers(0, 5)) * ib + (0 if ib == 1 else 0)
                        frame_bytes = rows * cols * ib
                        buf = rng.integers(0, 256, shift + n * (frame_bytes + gap) + 8, dtype=np.uint8)
                        offs = [shift + k * (frame_bytes + gap) for k in range(n)]
                        dt = np.dtype((">" if big else "<") + ("i" if rep else "u") + str(ib))
                        want = np.stack([np.frombuffer(buf[o:o + frame_bytes].tobytes(), dt).reshape(rows, cols) for o in offs])
                        want = want.astype(dt.newbyteorder("="))
                        fixed = want
                        if stored < bits:
                            fixed = (np.right_shift(np.left_shift(want, bits - stored), bits - stored) if rep
                                     else want & want.dtype.type((1 << stored) - 1))
                        for fix, ref in ((False, want), (True, fixed)):
                            if ib * rows * cols % 4 and n > 1:
                                continue
                            kw = dict(rows=rows, cols=cols, bits_allocated=bits, bits_stored=stored, pixel_representation=rep,
                                      big_endian=big, correct_unused_bits=fix, device=dev)
                            got = dicom._to_numpy(dicom._check_status(dicom.decode_frames(buf, offs, out="container", **kw)))
                            assert got.dtype == ref.dtype and np.array_equal(got, ref), (rows, cols, bits, rep, big, shift, fix)
                            g32 = dicom.decode_frames(buf, offs, out="float32", **kw).cpu().numpy()
                            assert np.array_equal(g32, ref.astype(np.float32))
                            g64 = dicom.decode_frames(buf, offs, out="float64", rescale=(1.25, -1000.5), **kw).cpu().numpy()
                            r64 = ref.astype(np.float64) * 1.25
                            r64 += -1000.5
                            assert np.array_equal(g64, r64)
    # a frame that pokes out of the buffer is reported, the others are decoded
    buf = rng.integers(0, 256, 1000, dtype=np.uint8)
    x = dicom.decode_frames(buf, [0, 900], rows=10, cols=10, bits_allocated=16, bits_stored=16, pixel_representation=0, device=dev)
    assert x._pl_status.cpu().tolist() == [0, 1]
    with pytest.raises(ValueError):
        dicom._check_status(x)
    assert np.array_equal(dicom._to_numpy(x)[0], np.frombuffer(buf[:200].tobytes(), "<u2").reshape(10, 10))


def check_ctp528_device_axis_path(dev, n_slices=8, size=256, mmpp=0.98, light=False):
    """Round 6: ct.ctp528_batch places the circle profiles about the centre line fitted ON THE DEVICE (pl_phantom_axis_fit)
    and reports the host's np.polyfit.  Held here: (1) the result equals the classic path's (the same call with the reported
    fits handed in: host centres, no device fit) bit for bit; (2) the device's placement fit agrees with np.polyfit to 1e-9
    and its median / np.isclose screen drops the same outlier slices; (3) when the margin test refuses some profiles
    (forced), those are sampled again about the exact centres and the result does not change; (4) a slice that does not
    show the phantom is left out of the fit exactly like the reference's `is_phantom_in_view` screen."""
    from unittest import mock

    from pylinac_amd import ct, ops
    from pylinac_amd.synthetic import catphan_volume

    vols = np.stack([catphan_volume(4000 + v, n_slices=n_slices, size=size, mm_per_pixel=mmpp) for v in range(2)])
    vols[1, 2] = vols[1, 2, 0, 0]                       # one slice of air: no phantom in view (status != 0)
    vols[1, 5] = np.roll(vols[1, 5], 9, axis=1)         # one slice shifted by 9 px: fails the isclose screen (atol 3)
    x = torch.from_numpy(vols).to(dev)
    res = ct.ctp528_batch(x, mmpp)
    classic = ct.ctp528_batch(x, mmpp, fit_zx=res["fit_zx"], fit_zy=res["fit_zy"])
    for k in ("rmtf", "maxs", "mins", "nregions", "center", "slices"):
        assert np.array_equal(res[k], classic[k], equal_nan=True), k
    assert torch.equal(torch.nan_to_num(res["profiles"], nan=-1.0), torch.nan_to_num(classic["profiles"], nan=-1.0))
    assert res["roi"][n_slices + 2, 0] != 0 and (res["roi"][:n_slices, 0] == 0).all()
    # the placement fit against the exact one
    roi_dev = torch.from_numpy(res["roi"]).to(dev)
    fit, cen, flag = ops.phantom_axis_fit(roi_dev, 2)
    fit, cen, flag = fit.cpu().numpy(), cen.cpu().numpy(), flag.cpu().numpy()
    assert (flag == 0).all()
    assert np.allclose(fit[:, :2], res["fit_zx"], rtol=1e-9, atol=1e-9) and np.allclose(fit[:, 2:], res["fit_zy"], rtol=1e-9, atol=1e-9)
    assert np.allclose(cen, res["center"], rtol=0, atol=1e-9)
    # (the shifted slice moved the fit if it had been kept: the exact fit without the screen differs)
    z = np.arange(n_slices)
    keep = np.ones(n_slices, bool)
    keep[2] = False
    loose = np.polyfit(z[keep], res["roi"][n_slices:, 4][keep], 1)
    assert abs(loose[1] - res["fit_zx"][1][1]) > 0.1
    # no phantom anywhere / one usable slice -> flags
    empty = res["roi"].copy()
    empty[:n_slices, 0] = 2
    single = res["roi"].copy()
    single[1:n_slices, 0] = 2
    f1 = ops.phantom_axis_fit(torch.from_numpy(empty).to(dev), 2)[2].cpu().numpy()
    f2 = ops.phantom_axis_fit(torch.from_numpy(single).to(dev), 2)[2].cpu().numpy()
    assert f1.tolist() == [1, 0] and f2.tolist() == [2, 0]
    # forced refusal of three profiles: sampled again about the exact centres, same answer
    real = ct._device_centres_disagree
    calls = []

    def refuse(*a, **k):
        assert real(*a, **k) is None
        calls.append(1)
        return np.array([0, 3, n_slices + 1])

    with mock.patch.object(ct, "_device_centres_disagree", refuse):
        again = ct.ctp528_batch(x, mmpp)
    assert calls
    for k in ("rmtf", "maxs", "mins", "nregions", "center"):
        assert np.array_equal(res[k], again[k], equal_nan=True), k
    assert torch.equal(torch.nan_to_num(res["profiles"], nan=-1.0), torch.nan_to_num(again["profiles"], nan=-1.0))
    # a subset of slices, one volume per chunk
    if not light:
        pick = np.array([1, n_slices + 3, n_slices + 4])
        part = ct.ctp528_batch(x, mmpp, slices=pick, chunk_volumes=1)
        assert np.array_equal(part["rmtf"], res["rmtf"][pick], equal_nan=True) and np.array_equal(part["center"], res["center"][pick])
    # the margin really is the distance to the nearest decision: moving a centre by less keeps every sample
    prof, idx, margin = ct.ctp528_profiles_batch(x[0], mmpp, None, None, slices_per_volume=n_slices,
                                                 device_centers=torch.from_numpy(res["center"][:n_slices]).to(dev))
    m = margin.cpu().numpy()
    assert (m > 0).all() and (m <= 0.5).all()
    for sign in (-1.0, 1.0):
        moved = res["center"][:n_slices] + sign * 0.5 * m[:, None]
        p2, _, _ = ct.ctp528_profiles_batch(x[0], mmpp, None, None, slices_per_volume=n_slices,
                                            device_centers=torch.from_numpy(moved).to(dev))
        assert torch.equal(torch.nan_to_num(p2, nan=-1.0), torch.nan_to_num(prof, nan=-1.0))


def check_edge_plane32(dev, shapes=((2, 70, 130, np.int16), (1, 40, 66, np.uint16), (2, 64, 256, np.int16), (1, 33, 300, np.uint16)),
                       sigmas=(1,), catphan_slices=()):
    """The packed-float32 edge plane (pl_edge_plane32) against the exact one (pl_edge_plane, pinned to scikit-image / scipy):
      * every stored value within `bracket` float32 bit patterns of RN32(exact) -- measured, not assumed --, an exact 0 stored
        as 0, and nothing else stored as 0;
      * the extrema over row spans and the raw maximum EXACT (bit for bit the float64 values of the exact path) for every
        slice whose status is 0, and status 0 for all but deliberately tied inputs;
      * the consumers on that plane with the bracket: Otsu threshold, all 256 histogram counts, the thresholded mask for
        thresholds ON plane values, the region table and the phantom ROI -- identical to the exact path's."""
    import torch

    from pylinac_amd import ct, ops

    rng = np.random.default_rng(23)
    worst = 0
    cases = []
    for n, h, w, dt in shapes:
        a = rng.integers(-1000 if dt == np.int16 else 0, 3000, (n, h, w)).astype(dt)
        a[:, h // 4:h // 2, w // 4:w // 2] += 700
        a[0, : h // 8] = 5                                        # a flat band: edge value exactly 0 there
        c0 = rng.integers(0, w // 2, h)
        c1 = np.minimum(c0 + rng.integers(1, w, h), w)
        cases.append((a, np.stack([c0, c1], 1).astype(np.int32)))
    if len(catphan_slices):
        from pylinac_amd.synthetic import catphan_volume

        vol = catphan_volume(seed=4000, n_slices=80)
        spans = ct._disk_spans_on_device(512, 512, 0.5, dev).cpu().numpy()
        cases.append((np.ascontiguousarray(vol[list(catphan_slices)]), spans))
    for a, spans_np in cases:
        x = torch.from_numpy(a).to(dev)
        spans = torch.from_numpy(spans_np).to(dev)
        n, h, w = a.shape
        for sigma in sigmas:
            e64, rawmax64, lo64, hi64 = ops.edge_plane(x, sigma, spans=spans, dtype=torch.float64)
            p32, rawmax, lo, hi, status, B = ops.edge_plane32(x, sigma, spans=spans)
            assert B >= 1 and not bool(status.any()), (a.shape, status.cpu().tolist())
            want32 = e64.to(torch.float32)
            d = (p32.view(torch.int32).to(torch.int64) - want32.view(torch.int32).to(torch.int64)).abs()
            worst = max(worst, int(d.max()))
            assert int(d.max()) <= B, (a.shape, sigma, int(d.max()))
            assert torch.equal(p32 == 0, e64 == 0)
            assert torch.equal(lo, lo64) and torch.equal(hi, hi64), (a.shape, sigma)
            small = rawmax64 < 32
            assert torch.equal(rawmax[small], rawmax64[small]) and torch.allclose(rawmax, rawmax64, rtol=1e-6, atol=0)
            # consumers
            thr64, raw64, work64 = ops.edge_otsu(e64, lo64, hi64, spans=spans, scale=0.8, return_work=True)
            thr, raw, work = ops.edge_otsu(p32, lo, hi, frames=x, sigma=sigma, spans=spans, scale=0.8, return_work=True, bracket=B)
            assert torch.equal(thr, thr64) and torch.equal(raw, raw64) and torch.equal(work[:, :256], work64[:, :256]), (a.shape, sigma)
            if ops.mask_regions_fits(h, w, 64) and min(h, w) >= 12:
                # thresholds ON plane values (an exact float64 value; a float32 value): the pixels around them are undecided
                # within the bracket and take the exact recomputation.  (High ones: a mask of few row runs fits the LDS list.)
                k_hi = max(h * w // 500, 8)
                for tv in (thr64, e64.reshape(n, -1).sort(dim=1).values[:, -k_hi].contiguous(),
                           want32.reshape(n, -1).sort(dim=1).values[:, -2 * k_hi].to(torch.float64).contiguous(),
                           torch.full_like(thr64, -1.0), torch.full_like(thr64, float("inf"))):
                    q = ops.edge_regions(p32, x, sigma, tv.contiguous(), 0, False, 64, return_mask=True, want_table=False, bracket=B)
                    ok = q["status"] == 0                                     # (status 1 = more row runs than the LDS list holds)
                    assert bool(ok.any()), (a.shape, sigma)
                    assert torch.equal(q["mask"][ok], (e64 > tv[:, None, None]).to(torch.uint8)[ok]), (a.shape, sigma)
                r64 = ops.mask_regions(e64, thr64, 2, True, 64)
                r32 = ops.edge_regions(p32, x, sigma, thr64, 2, True, 64, bracket=B)
                assert torch.equal(r32["status"], r64[2]) and torch.equal(r32["count"], r64[1]) and torch.equal(r32["table"], r64[0])
    # ties at the extremum: two pixels of the selection with the SAME smallest value in one lane's columns cannot be ranked
    # by the float32 plane -- the slice must say so (status 1), not guess
    t = np.zeros((1, 48, 64), np.int16)
    t[0, :, 32:] = 1000                                            # a vertical step: every row has the same edge profile
    x = torch.from_numpy(t).to(dev)
    sp = torch.from_numpy(np.tile(np.array([[24, 40]], np.int32), (48, 1))).to(dev)
    _, _, lo_t, hi_t, st_t, _ = ops.edge_plane32(x, 1, spans=sp)
    _, _, lo_e, hi_e = ops.edge_plane(x, 1, spans=sp, dtype=torch.float64)
    assert int(st_t[0]) == 1 or (torch.equal(lo_t, lo_e) and torch.equal(hi_t, hi_e))
    # ... and the localisation repeats such a slice on the exact path: the ROI table is the same with the knob on and off
    both = np.concatenate([t, rng.integers(-1000, 1000, (1, 48, 64)).astype(np.int16)])
    xb = torch.from_numpy(both).to(dev)
    on = ct.phantom_roi_batch(xb, 4.0)
    try:
        ct.EDGE_PLANE32 = False
        off = ct.phantom_roi_batch(xb, 4.0)
    finally:
        ct.EDGE_PLANE32 = True
    assert np.array_equal(on, off, equal_nan=True), (on, off)
    return worst
