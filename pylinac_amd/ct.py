"""Batched CatPhan slice localisation.

Mirrors ``get_regions`` (pylinac/ct.py:3315-3348, Slice branch, default
``clip_in_localization=False`` ct.py:2043) and the region choice of ``Slice.phantom_roi``
(ct.py:381-425) for every slice of a device-resident int16/uint16/float batch:

    edges = filters.scharr(array.astype(float))                 -> pl_scharr
    edges = filters.gaussian(edges, sigma=1)                    -> pl_gaussian2d_mode('nearest')
    thres = threshold_otsu(edges[disk(center, 110 mm)]) * 0.8   -> pl_minmax/pl_hist_uniform + 256-bin Otsu
    bw = edges > thres                                          -> pl_compare
    bw = clear_border(bw, buffer_size=min(int(max(shape)/100), 3))   -> pl_clear_border
    bw = binary_fill_holes(bw)                                  -> pl_fill_holes (4-connected background)
    labeled, num = measure.label(bw)                            -> pl_label (8-connected)
    regionprops(labeled, edges)                                 -> pl_region_stats
    phantom = argmin |filled_area - catphan_size|, 1.3x window  -> 256-bin / per-region scalars on the host

The per-slice scalars (256-bin Otsu on the counts, picking the region) are a few hundred flops and
run in numpy on the host, like the other per-dataset glue of the reference.
``filled_area`` equals ``area`` here because the regions come out of ``binary_fill_holes`` (a
4-connected-background fill leaves no 8-connected-background hole); that identity is asserted
against scikit-image on the golden slices.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from . import ops, regionprops as _rp

CATPHAN_RADIUS_MM = 101  # pylinac/ct.py: CatPhanBase.catphan_radius_mm
EDGE_PLANE32 = os.environ.get("PL_EDGE_PLANE32", "1") != "0"   # the localisation's edge image from pl_edge_plane32 (0: the exact
                                                             # float64 kernel -- an A/B knob; results are identical)


def disk_mask(center_rc, radius: float, shape) -> np.ndarray:
    """``skimage.draw.disk(center, radius, shape=shape)`` as a uint8 mask (draw.py ellipse +
    _ellipse_in_shape with rotation 0): pure index geometry, evaluated once per slice shape."""
    center = np.array(center_rc, dtype=float)
    radii = np.array([radius, radius], dtype=float)
    upper_left = np.maximum(np.ceil(center - radii).astype(int), 0)
    lower_right = np.minimum(np.floor(center + radii).astype(int), np.array(shape[:2]) - 1)
    shifted = center - upper_left
    bshape = lower_right - upper_left + 1
    r_lim, c_lim = np.ogrid[0:float(bshape[0]), 0:float(bshape[1])]
    r, c = (r_lim - shifted[0]), (c_lim - shifted[1])
    dist = ((r * 1.0 + c * 0.0) / radii[0]) ** 2 + ((r * 0.0 - c * 1.0) / radii[1]) ** 2
    rr, cc = np.nonzero(dist < 1)
    m = np.zeros(shape, np.uint8)
    m[rr + upper_left[0], cc + upper_left[1]] = 1
    return m


_DISK_CACHE: dict = {}


def row_spans(mask: np.ndarray) -> np.ndarray | None:
    """int32 [H, 2] column intervals [c0, c1) of a mask whose rows are single runs (a disk); None when some row is not."""
    m = np.asarray(mask) != 0
    h, w = m.shape
    cnt = m.sum(axis=1)
    c0 = np.where(cnt > 0, m.argmax(axis=1), 0)
    c1 = c0 + cnt
    ok = all(m[r, c0[r]:c1[r]].all() for r in range(h))
    return np.stack([c0, c1], axis=1).astype(np.int32) if ok else None


def _disk_spans_on_device(h: int, w: int, mm_per_pixel: float, dev) -> torch.Tensor:
    key = ("spans", h, w, float(mm_per_pixel), str(dev))
    if key not in _DISK_CACHE:
        cy, cx = h / 2 - 0.5, w / 2 - 0.5
        _DISK_CACHE[key] = torch.from_numpy(row_spans(disk_mask((cy, cx), 110 / mm_per_pixel, (h, w)))).to(dev)
    return _DISK_CACHE[key]


def STAGE_TIMERS(x: torch.Tensor, mm_per_pixel: float):
    """(name, thunk) pairs for scripts/time_ct_stages.py: the entry points of the localisation, each on its own."""
    n, h, w = x.shape
    disk = _disk_on_device(h, w, mm_per_pixel, x.device)
    spans = _disk_spans_on_device(h, w, mm_per_pixel, x.device)
    yield "edge_plane float32 + spans", lambda: ops.edge_plane(x, 1, spans=spans)
    yield "edge_plane float64 + spans", lambda: ops.edge_plane(x, 1, spans=spans, dtype=torch.float64)
    yield "edge_plane float32 + byte mask", lambda: ops.edge_plane(x, 1, mask=disk)
    yield "edge_plane extrema only", lambda: ops.edge_plane(x, 1, spans=spans, want_plane=False)
    e64, _, lo, hi = ops.edge_plane(x, 1, spans=spans, dtype=torch.float64)
    yield "otsu_float_masked (float64 plane, round 3)", lambda: ops.otsu_float_masked(e64, disk, scale=0.8, lohi=(lo, hi))
    thr, _ = ops.otsu_float_masked(e64, disk, scale=0.8, lohi=(lo, hi))
    yield "mask_regions (float64 plane, round 3)", lambda: ops.mask_regions(e64, thr, min(int(max(h, w) / 100), 3) + 1, True, 64)
    p32, rawmax, lo, hi = ops.edge_plane(x, 1, spans=spans)
    yield "edge_otsu (float32 plane, one launch)", lambda: ops.edge_otsu(p32, lo, hi, frames=x, sigma=1, spans=spans, scale=0.8)
    yield "edge_otsu (float64 plane, one launch)", lambda: ops.edge_otsu(e64, lo, hi, spans=spans, scale=0.8)
    yield "edge_regions (float32 plane + ROI choice)", lambda: ops.edge_regions(
        p32, x, 1, thr, min(int(max(h, w) / 100), 3) + 1, True, 64, catphan_size=np.pi * 101**2 / mm_per_pixel**2, rawmax=rawmax,
        want_table=False)
    yield "phantom_roi_batch", lambda: phantom_roi_batch(x, mm_per_pixel)


def _disk_on_device(h: int, w: int, mm_per_pixel: float, dev) -> torch.Tensor:
    key = (h, w, float(mm_per_pixel), str(dev))
    if key not in _DISK_CACHE:
        cy, cx = h / 2 - 0.5, w / 2 - 0.5                                  # BaseImage.center, image.py:527-533
        _DISK_CACHE[key] = torch.from_numpy(disk_mask((cy, cx), 110 / mm_per_pixel, (h, w))).to(dev)
    return _DISK_CACHE[key]


def get_regions_batch(slices: torch.Tensor, mm_per_pixel: float, fill_holes: bool = True,
                      clear_borders: bool = True, max_labels: int = 64, raw_edges: torch.Tensor | None = None):
    """-> dict(edges f64 [N,H,W], bw u8, labels i32, num i32 [N], stats f64 [N,max_labels,10],
    overflow i32 [N], otsu f64 [N] (device)).  Nothing returns to the host: the 256-bin Otsu of the disk-masked edge
    image (np.linspace edges, np.histogram binning, skimage's class statistics) runs in kernels too.
    ``raw_edges``: ``ops.scharr(slices)`` when the caller already has it."""
    x = ops._frames(slices)
    n, h, w = x.shape
    dev = x.device
    edges = ops.gaussian_filter_mode(ops.scharr(x) if raw_edges is None else raw_edges, 1, "nearest")
    disk = _disk_on_device(h, w, mm_per_pixel, dev)
    # np.histogram(edges[disk], 256): the range is the min/max of the SELECTED pixels
    thr, otsu = ops.otsu_float_masked(edges, disk, scale=0.8)
    bw = ops.compare(edges, thr, ">")
    if clear_borders:
        bw = ops.clear_border(bw, min(int(max(h, w) / 100), 3))
    if fill_holes:
        bw = ops.fill_holes(bw, 4)
    labels, num = ops.label(bw, 8)
    stats, ovf = ops.region_stats(labels, edges, max_labels)
    return dict(edges=edges, bw=bw, labels=labels, num=num, stats=stats, overflow=ovf, otsu=otsu)


def _select_phantom_roi(stats: np.ndarray, num: np.ndarray, ovf: np.ndarray, raw_max: np.ndarray, catphan_size: float,
                        max_labels: int) -> np.ndarray:
    """The choice of ``Slice.phantom_roi`` (``sorted(regionprops, key=|filled_area - catphan_size|)[0]`` and the size test,
    ct.py:398-409) from region tables on the host, vectorised over slices: the general path's half of ``phantom_roi_batch``
    (the fused path makes the same choice inside ``pl_edge_regions``)."""
    n = stats.shape[0]
    valid = np.arange(max_labels)[None, :] < num[:, None]
    filled = stats[:, :, 0]                                 # == filled_area after binary_fill_holes
    dist = np.where(valid, np.abs(filled - catphan_size), np.inf)
    k = np.argmin(dist, axis=1)                             # first minimum = the stable sort's first element
    rows = np.arange(n)
    t = stats[rows, k]
    fk = t[:, 0]
    status = np.zeros(n)
    status[(catphan_size * 1.3 < fk) | (fk < catphan_size / 1.3)] = 3
    status[ovf != 0] = 4
    status[num < 1] = 2
    status[raw_max < 0.1] = 1
    out = np.full((n, 8), np.nan)
    out[:, 0] = status
    ok = status == 0
    with np.errstate(invalid="ignore", divide="ignore"):
        good = np.stack([np.zeros(n), k + 1.0, fk, t[:, 5] / t[:, 0], t[:, 6] / t[:, 0], t[:, 1], t[:, 2], t[:, 3]], axis=1)
    out[ok] = good[ok]
    return out


def _phantom_roi_general(x: torch.Tensor, mm_per_pixel: float, catphan_size: float, max_labels: int) -> np.ndarray:
    raw = ops.scharr(x)                                              # computed once: the edge test and get_regions
    raw_max_t = ops.minmax(raw)[1]                                   # ct.py:392: np.max(edges) < 0.1
    reg = get_regions_batch(x, mm_per_pixel, fill_holes=True, clear_borders=True, max_labels=max_labels, raw_edges=raw)
    return _select_phantom_roi(reg["stats"].cpu().numpy(), np.minimum(reg["num"].cpu().numpy(), max_labels),
                               reg["overflow"].cpu().numpy(), raw_max_t.cpu().numpy(), catphan_size, max_labels)


def phantom_roi_batch(slices: torch.Tensor, mm_per_pixel: float, catphan_radius_mm: float = CATPHAN_RADIUS_MM,
                      max_labels: int = 64) -> np.ndarray:
    """``Slice.phantom_roi`` for every slice -> float64 [N, 8]:
    columns: status (0 ok, 1 no edges, 2 no ROI, 3 wrong size, 4 label overflow), label,
    filled_area, centroid_r, centroid_c, bbox_r0, bbox_c0, bbox_r1 -- the reference raises ValueError
    for status 1-3; the batch reports per-slice codes instead.

    int16 / uint16 slices take THREE launches and ONE transfer of 64 bytes per slice: ``pl_edge_plane`` (smoothed Scharr
    plane as float32 + its exact extrema on the 110 mm disk), ``pl_edge_otsu`` (the disk histogram and its Otsu threshold)
    and ``pl_edge_regions`` (threshold, clear_border, fill_holes, label, regionprops and the choice of the phantom region,
    one workgroup per slice)."""
    return _phantom_roi_finish(_phantom_roi_launch(slices, mm_per_pixel, catphan_radius_mm, max_labels))


def _phantom_roi_launch(slices: torch.Tensor, mm_per_pixel: float, catphan_radius_mm: float = CATPHAN_RADIUS_MM,
                        max_labels: int = 64):
    """The device half of ``phantom_roi_batch``: queues the three launches and the transfer of the ROI table, returns at once."""
    x = ops._frames(slices)
    n, h, w = x.shape
    catphan_size = np.pi * catphan_radius_mm**2 / mm_per_pixel**2        # ct.py:2581-2584
    args = (x, mm_per_pixel, catphan_size, max_labels)
    if not (x.dtype in (torch.int16, torch.uint16) and ops.mask_regions_fits(h, w, max_labels)):
        return args, None
    spans = _disk_spans_on_device(h, w, mm_per_pixel, x.device)
    if w % 2 == 0 and EDGE_PLANE32:
        # round 6: the edge image in packed float32 (half the vector instructions of the exact kernel); the histogram and the
        # threshold decide from it within its bracket and recompute exactly what they cannot decide, the extrema are exact.
        # A slice whose extrema the kernel could not certify (status 1: two tied candidates in one lane) is marked 5 in the
        # ROI table and repeated on the general path by _phantom_roi_finish, like a slice with too many row runs
        plane, raw_max, lo, hi, unsure, bracket = ops.edge_plane32(x, 1, spans=spans)
    else:
        plane, raw_max, lo, hi = ops.edge_plane(x, 1, spans=spans)
        unsure, bracket = None, 1
    thr, _ = ops.edge_otsu(plane, lo, hi, frames=x, sigma=1, spans=spans, scale=0.8, bracket=bracket)
    reg = ops.edge_regions(plane, x, 1, thr, min(int(max(h, w) / 100), 3) + 1, True, max_labels, catphan_size=catphan_size,
                           rawmax=raw_max, want_table=False, bracket=bracket)
    if unsure is not None:
        reg["roi"][:, 0] = torch.where(unsure != 0, torch.full_like(reg["roi"][:, 0], 5.0), reg["roi"][:, 0])
    copy = ops.HostCopy(reg["roi"])
    copy.device_table = reg["roi"]                                       # (ctp528_batch goes on from it without waiting)
    return args, copy


def _phantom_roi_finish(pending) -> np.ndarray:
    (x, mm_per_pixel, catphan_size, max_labels), copy = pending
    if copy is None:
        return _phantom_roi_general(x, mm_per_pixel, catphan_size, max_labels)
    out = copy.numpy().copy()                                            # the one synchronisation of the localisation
    redo = np.flatnonzero(out[:, 0] == 5)                                # slices with more row runs than the LDS list holds
    if len(redo):
        sub = x[torch.from_numpy(redo).to(x.device)].contiguous()
        out[redo] = _phantom_roi_general(sub, mm_per_pixel, catphan_size, max_labels)
    return out


# ----------------------------------------------------------------------------------------------------------
# Volume-level localisation (pylinac/ct.py:2398-2508): the loop over every slice of BASELINE config #5
# ----------------------------------------------------------------------------------------------------------
def find_phantom_axis_volume(slices: torch.Tensor, mm_per_pixel: float, catphan_radius_mm: float = CATPHAN_RADIUS_MM,
                             x_adjustment: float = 0, y_adjustment: float = 0, roi: np.ndarray | None = None):
    """``CatPhanBase.find_phantom_axis`` (ct.py:2398-2446) for a resident volume [Z, H, W]: the phantom ROI of EVERY
    slice in one batch, then the reference's outlier screen (``np.isclose`` to the median, atol 3, rtol 0.01) and the
    two first-order fits of the centre against z.  -> (fit_zx coefficients, fit_zy coefficients, roi table)."""
    if roi is None:
        roi = phantom_roi_batch(slices, mm_per_pixel, catphan_radius_mm)
    seen = roi[:, 0] == 0                                   # is_phantom_in_view()
    zs = np.flatnonzero(seen)
    if len(zs) == 0:
        raise ValueError("The phantom was not found in any slice")
    center_xs = roi[seen, 4] + x_adjustment
    center_ys = roi[seen, 3] + y_adjustment
    x_idxs = np.argwhere(np.isclose(np.median(center_xs), center_xs, atol=3, rtol=0.01))
    y_idxs = np.argwhere(np.isclose(np.median(center_ys), center_ys, atol=3, rtol=0.01))
    common = np.intersect1d(x_idxs, y_idxs)
    fit_zx = np.polyfit(zs[common], center_xs[common], deg=1, rcond=0.00001)
    fit_zy = np.polyfit(zs[common], center_ys[common], deg=1, rcond=0.00001)
    return fit_zx, fit_zy, roi


def _polyfit1_stack(xs: np.ndarray, ys: np.ndarray):
    """``np.polyfit(x_g, y_g, deg=1, rcond=0.00001)`` for G data sets of the same length at once, bit-identical to G separate
    calls: polyfit's own steps (Vandermonde columns scaled to unit norm, LAPACK ``gelsd``, rescaling;
    numpy/lib/_polynomial_impl.py) with the solver's stacked form -- the gufunc behind ``np.linalg.lstsq`` solves every
    leading-dimension item on its own with a single right-hand side, exactly like a separate call.  -> [G, 2] or None when
    that gufunc is not there (the caller then loops)."""
    if not _polyfit_stack_usable():
        return None
    try:
        return _polyfit1_stack_raw(xs, ys)
    except Exception:                                        # a numpy whose private gufunc differs: the caller loops
        return None


_POLYFIT_STACK_OK: bool | None = None


def _polyfit_stack_usable() -> bool:
    """The stacked solver goes through a PRIVATE numpy gufunc with a hard-coded signature and mirrors np.polyfit's internal
    steps; it is trusted only after it has reproduced np.polyfit bit for bit on a small sample in this process (once)."""
    global _POLYFIT_STACK_OK
    if _POLYFIT_STACK_OK is None:
        try:
            rng = np.random.default_rng(12345)
            xs = np.sort(rng.uniform(0, 80, (6, 17)), axis=1)
            ys = 255.5 + 0.01 * xs + rng.normal(0, 0.3, xs.shape)
            got = _polyfit1_stack_raw(xs, ys)
            want = np.stack([np.polyfit(x, y, deg=1, rcond=0.00001) for x, y in zip(xs, ys)])
            _POLYFIT_STACK_OK = got is not None and np.array_equal(got, want)
        except Exception:
            _POLYFIT_STACK_OK = False
    return _POLYFIT_STACK_OK


def _polyfit1_stack_raw(xs: np.ndarray, ys: np.ndarray):
    from numpy.linalg import _umath_linalg as _ul

    gufunc = _ul.lstsq
    lhs = np.stack([xs, np.ones_like(xs)], axis=2).astype(np.float64)          # np.vander(x, 2)
    scale = np.sqrt((lhs * lhs).sum(axis=1))
    lhs = lhs / scale[:, None, :]
    with np.errstate(invalid="ignore", over="ignore", divide="ignore", under="ignore"):
        c, _, rank, _ = gufunc(lhs, ys[:, :, None].astype(np.float64), 0.00001, signature="ddd->ddid")
    if (rank != 2).any():
        return None
    return c[:, :, 0] / scale


def find_phantom_axes_batch(roi: np.ndarray, n_volumes: int, x_adjustment: float = 0, y_adjustment: float = 0):
    """``find_phantom_axis_volume`` for the ROI table of ``n_volumes`` equally long volumes -> (fit_zx [V, 2], fit_zy [V, 2]).
    When every slice of every volume shows the phantom (the usual case) the medians, the outlier screen and the two
    first-order fits of all volumes are taken together; the fits stay bit-identical to ``np.polyfit`` per volume
    (``_polyfit1_stack``).  Anything else goes volume by volume."""
    spv = roi.shape[0] // n_volumes
    r = roi.reshape(n_volumes, spv, roi.shape[1])

    def loop():
        fits = [find_phantom_axis_volume(None, 0.0, roi=r[v], x_adjustment=x_adjustment, y_adjustment=y_adjustment)
                for v in range(n_volumes)]
        return np.stack([f[0] for f in fits]), np.stack([f[1] for f in fits])

    if not (r[:, :, 0] == 0).all():
        return loop()
    cxs, cys = r[:, :, 4] + x_adjustment, r[:, :, 3] + y_adjustment
    okx = np.isclose(np.median(cxs, axis=1)[:, None], cxs, atol=3, rtol=0.01)
    oky = np.isclose(np.median(cys, axis=1)[:, None], cys, atol=3, rtol=0.01)
    common = okx & oky
    cnt = common.sum(axis=1)
    if cnt.min() < 2:
        return loop()
    fzx = np.empty((n_volumes, 2))
    fzy = np.empty((n_volumes, 2))
    zs = np.arange(spv, dtype=np.float64)
    for m in np.unique(cnt):                                   # volumes with the same number of kept slices together
        g = np.flatnonzero(cnt == m)
        sel = common[g]
        xs = np.broadcast_to(zs, sel.shape)[sel].reshape(len(g), m)
        cx = _polyfit1_stack(xs, cxs[g][sel].reshape(len(g), m))
        cy = _polyfit1_stack(xs, cys[g][sel].reshape(len(g), m))
        if cx is None or cy is None:
            return loop()
        fzx[g], fzy[g] = cx, cy
    return fzx, fzy


def find_origin_slice_volume(slices: torch.Tensor, mm_per_pixel: float, fit_zx, fit_zy, slice_thickness: float,
                             localization_radius: float = 59, hu_origin_slice_variance: float = 400,
                             catphan_radius_mm: float = CATPHAN_RADIUS_MM, roi: np.ndarray | None = None) -> int:
    """``CatPhanBase.find_origin_slice`` (ct.py:2453-2508): for every second slice that shows the phantom, a collapsed
    circle profile (5 radii, +-5 %) through the HU inserts about the fitted centre, and the percentile / median test for
    "both very low and very high HU, little variation in between"; the median of the qualifying slices is the centre of
    the HU module.  Profiles and their order statistics are one batch on the device."""
    from .canny import _percentile_f64

    x = ops._frames(slices)
    n, h, w = x.shape
    idx = np.arange(0, n, 2)
    if roi is None:
        roi = phantom_roi_batch(x, mm_per_pixel, catphan_radius_mm)
    idx = idx[roi[idx, 0] == 0]
    if len(idx) == 0:
        raise ValueError("No slices were found that resembled the HU linearity module")
    cx = np.polyval(fit_zx, idx)
    cy = np.polyval(fit_zy, idx)
    radius = localization_radius / mm_per_pixel
    if (w < radius + cx).any() or (h < radius + cy).any():        # CircleProfile._ensure_array_size (profile.py:2394-2402)
        raise ValueError("Array size not large enough to compute profile")
    radii = np.linspace(radius * 0.95, radius * 1.05, 5)
    sub = x[torch.from_numpy(idx).to(x.device)].contiguous()
    prof = ops.circle_profile(sub, cx, cy, radii, np.pi * radii.max() * 2, 0, True, 5.0)       # [M, L] float64
    p = prof[:, None, :].contiguous()                                                       # one "frame" per profile
    low_end, high_end, p80, p20, median = (_percentile_f64(p, q).cpu().numpy() for q in (2, 98, 80, 20, 50))
    variation_limit = max(100, slice_thickness * -100 + 300)
    hit = ((low_end < median - hu_origin_slice_variance) & (high_end > median + hu_origin_slice_variance)
           & ((p80 - p20) < variation_limit))
    hu_slices = idx[hit]
    if len(hu_slices) == 0:
        raise ValueError("No slices were found that resembled the HU linearity module")
    c = int(round(float(np.median(hu_slices))))
    ln = len(hu_slices)
    hu_slices = hu_slices[((c + ln / 2) >= hu_slices) & (hu_slices >= (c - ln / 2))]
    center = int(round(float(np.median(hu_slices))))
    return center if 0 <= center < n else None


def find_phantom_roll_volume(slices: torch.Tensor, mm_per_pixel: float, origin_slice: int, fit_zx,
                             air_bubble_radius_mm: float = 7, slice_offset: int = 0) -> float:
    """``CatPhanBase.find_phantom_roll`` (ct.py:2517-2563): the regions of the HU slice (``get_regions`` without hole
    filling) that look like the air bubbles -- ``filled_area`` within a factor 2 of the bubble disk and
    ``eccentricity`` < 0.5 (regionprops semantics: holes filled inside the region's own bounding box with the full
    structuring element; eccentricity from the eigenvalues of the region's inertia tensor) -- the two closest to the
    phantom's x centre, and the angle of the line through them.  Labelling, region table and the per-candidate hole
    filling run on the device; a handful of candidates are screened on the host."""
    import math
    import warnings

    x = ops._frames(slices)
    k = int(origin_slice) + int(slice_offset)
    reg = get_regions_batch(x[k:k + 1], mm_per_pixel, fill_holes=False, clear_borders=True, max_labels=256)
    if int(reg["overflow"][0]):
        raise RuntimeError("more than 256 regions in the roll slice")
    num = int(reg["num"][0])
    stats = reg["stats"][0, :num].cpu().numpy()
    labels = reg["labels"][0]
    thresh = np.pi * ((air_bubble_radius_mm / mm_per_pixel) ** 2)
    bubbles = []
    for j in range(num):
        area, r0, c0, r1, c1 = stats[j, 0], int(stats[j, 1]), int(stats[j, 2]), int(stats[j, 3]), int(stats[j, 4])
        # filled_area lies between the region's area and its bounding-box area: skip what cannot qualify
        if not (area < thresh * 2 and (r1 - r0) * (c1 - c0) > thresh / 2):
            continue
        crop = (labels[r0:r1, c0:c1] == j + 1).to(torch.uint8).contiguous()
        filled_area = float(ops.fill_holes(crop[None], connectivity_bg=8)[0].sum())
        if not thresh * 2 > filled_area > thresh / 2:
            continue
        mom, _ = ops.region_moments(crop.to(torch.int32)[None], 1)            # exact integer raw moments
        ecc = _rp.eccentricity(tuple(int(v) for v in mom[0, 0].cpu().tolist()))
        if ecc < 0.5:
            bubbles.append((stats[j, 5] / area, stats[j, 6] / area))          # centroid (row, col)
    cx = float(np.polyval(fit_zx, k))
    central = sorted(bubbles, key=lambda b: abs(b[1] - cx))[:2]
    top_bottom = sorted(central, key=lambda b: b[0])
    if len(top_bottom) < 2:
        warnings.warn("Could not determine phantom roll. Setting roll to 0.", UserWarning)
        return 0.0
    y_dist = top_bottom[1][0] - top_bottom[0][0]
    x_dist = top_bottom[1][1] - top_bottom[0][1]
    return float(np.rad2deg(np.arctan2(y_dist, x_dist)) - 90)


# ----------------------------------------------------------------------------------------------------------
# CTP528 spatial resolution per slice (pylinac/ct.py:1398-1580): BASELINE config #5's per-slice record
# ----------------------------------------------------------------------------------------------------------
# line-pair regions of the CatPhan 504 / 604 (pylinac/ct.py:1417-1503): start, end (fractions of the circle profile),
# number of peaks, number of valleys, peak spacing (fraction), lp/mm
CTP528_REGIONS = (
    (0, 0.107, 2, 1, 0.021, 0.1), (0.107, 0.173, 3, 2, 0.01, 0.2), (0.173, 0.236, 4, 3, 0.006, 0.3),
    (0.236, 0.286, 4, 3, 0.00557, 0.4), (0.286, 0.335, 4, 3, 0.004777, 0.5), (0.335, 0.387, 5, 4, 0.00398, 0.6),
    (0.387, 0.434, 5, 4, 0.00358, 0.7), (0.434, 0.479, 5, 4, 0.0027866, 0.8),
)


_BEYOND_CACHE: dict = {}


def ctp528_profiles_batch(volume: torch.Tensor, mm_per_pixel: float, fit_zx, fit_zy, slices=None, roll_deg: float = 0.0,
                          radius2linepairs_mm: float = 47, scaling_factor: float = 1.0, roi_size_factor: float = 1.0,
                          start_angle: float = np.pi, ccw: bool = True, slices_plusminus: int = 3,
                          slices_per_volume: int | None = None, device_centers: torch.Tensor | None = None):
    """``CTP528CP504.circle_profile`` (pylinac/ct.py:1559-1580) for the chosen slices of resident volumes: a stack
    [S, H, W] of one volume, or of several volumes of ``slices_per_volume`` slices each (``fit_zx`` / ``fit_zy`` are then
    [V, 2] coefficient tables and ``slices`` indexes the stack).  ``combine_surrounding_slices(+-3, "max")`` (module
    attributes ``combine_method = "max"``, ``num_slices = 3``, ct.py:1415-1416) inside each slice's own volume, a
    CollapsedCircleProfile of 20 radii within +-4 % of the line-pair radius at 2x sampling about the phantom centre
    ``(fit_zx(z), fit_zy(z))``, ``filter(0.001, "gaussian")``, ``ground()``.
    -> (float64 [M, L] profiles on the device, the slice indices into the stack).
    ``device_centers`` (float64 [S, 2] on the device, ``ops.phantom_axis_fit``) instead of the fits: nothing comes from or
    goes to the host, the array-size test is left to the caller, and the result is (profiles, indices, margin [M])."""
    from .array_utils import resolve_filter_size

    x = ops._frames(volume)
    n, h, w = x.shape
    spv = int(slices_per_volume or n)
    idx = np.arange(n) if slices is None else np.asarray(slices, dtype=np.int64)
    v, z = idx // spv, idx % spv                                               # volume and slice number inside it
    radius = radius2linepairs_mm * scaling_factor / mm_per_pixel               # ct.py:1546-1549
    margin = None
    if device_centers is None:
        fzx, fzy = np.atleast_2d(np.asarray(fit_zx, dtype=np.float64)), np.atleast_2d(np.asarray(fit_zy, dtype=np.float64))
        cx = fzx[v, 0] * z + fzx[v, 1]                                         # np.poly1d(fit)(z), ct.py:434-439
        cy = fzy[v, 0] * z + fzy[v, 1]
        if (w < radius + cx).any() or (h < radius + cy).any():                 # CircleProfile._ensure_array_size
            raise ValueError("Array size not large enough to compute profile")
    else:
        cen = device_centers if slices is None else device_centers[_device_index(idx, x.device)]
        cx, cy = cen[:, 0].contiguous(), cen[:, 1].contiguous()
    width_ratio, num_profiles, sampling_ratio = 0.04 * roi_size_factor, 20, 2
    radii = np.linspace(radius * (1 - width_ratio), radius * (1 + width_ratio), num_profiles)   # profile.py:2448-2452
    size = np.pi * radii.max() * 2 * sampling_ratio
    if device_centers is not None:
        if not (x.dtype in (torch.int16, torch.uint16, torch.int32, torch.uint8) and len(idx) <= 65535):
            raise TypeError("device centres: integer slices, at most 65535 profiles")
        prof, margin = ops.circle_profile(x, cx, cy, radii, size, start_angle + np.deg2rad(roll_deg), ccw, float(num_profiles),
                                          combine=(idx, spv, slices_plusminus), want_margin=True)
    elif x.dtype in (torch.int16, torch.uint16, torch.int32, torch.uint8) and len(idx) <= 65535:
        # the +-3-slice maximum is taken per tap of the ring: the combined slices are never built
        prof = ops.circle_profile(x, cx, cy, radii, size, start_angle + np.deg2rad(roll_deg), ccw, float(num_profiles),
                                  combine=(idx, spv, slices_plusminus))
    else:
        combined = ops.combine_slices(x, slices_plusminus, "max", spv)
        sub = combined if slices is None else combined[torch.from_numpy(idx).to(x.device)].contiguous()
        prof = ops.circle_profile(sub, cx, cy, radii, size, start_angle + np.deg2rad(roll_deg), ccw, float(num_profiles))
    sigma = resolve_filter_size(prof.shape[1], 0.001)                          # array_utils.filter: int(round(len * size))
    prof = ops.gaussian_filter1d(prof, sigma, axis=-1)
    mn, _ = ops.minmax(prof[:, None, :])
    prof = ops.ground(prof[:, None, :].contiguous(), mn=mn)[:, 0, :].contiguous()
    # combine_surrounding_slices indexes dicomstack[z - k .. z + k] (ct.py:3375-3378): negative indices wrap around to the end
    # of the stack (reproduced by pl_combine_slices), indices past the last slice raise IndexError in the reference -- those
    # slices get a NaN profile here (-> no regions, NaN rMTF downstream)
    if bool((z + slices_plusminus >= spv).any()):
        key = ((z + slices_plusminus >= spv).tobytes(), str(prof.device))     # (the same mask pass after pass: no upload mid-pass)
        beyond = _BEYOND_CACHE.get(key)
        if beyond is None:
            if len(_BEYOND_CACHE) > 16:
                _BEYOND_CACHE.clear()
            beyond = _BEYOND_CACHE[key] = torch.from_numpy(z + slices_plusminus >= spv).to(prof.device)
        prof = prof.masked_fill(beyond[:, None], float("nan"))
    return (prof, idx) if device_centers is None else (prof, idx, margin)


_INDEX_CACHE: dict = {}


def _device_index(idx: np.ndarray, dev) -> torch.Tensor:
    key = (idx.tobytes(), str(dev))
    hit = _INDEX_CACHE.get(key)
    if hit is None:
        if len(_INDEX_CACHE) > 16:
            _INDEX_CACHE.clear()
        hit = _INDEX_CACHE[key] = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(dev)
    return hit


def ctp528_mtf_batch(profiles: torch.Tensor, regions=CTP528_REGIONS):
    """``CTP528CP504.mtf`` (pylinac/ct.py:1511-1544) for a batch of circle profiles [M, L]: per line-pair region the
    ``num peaks`` most prominent peaks (``find_peaks``, threshold 0.3, the region as search window), the valleys between
    the outermost of them (``find_valleys`` = peaks of the negated profile, per-profile search window), their means,
    Michelson contrast, normalised to region 1.  A slice stops at the first region with the wrong number of peaks, like
    the reference's ``break``.  -> dict(rmtf float64 [M, 8] (NaN beyond the regions found; all NaN = the reference's
    "Did not find any spatial resolution pairs"), nregions int [M], maxs / mins float64 [M, 8])."""
    return _ctp528_mtf_finish(_ctp528_mtf_launch(profiles, regions))


def _ctp528_mtf_launch(profiles: torch.Tensor, regions=CTP528_REGIONS):
    p = profiles.contiguous()
    m, length = p.shape
    # ONE launch for the sixteen searches of a profile (a wave per (profile, region) pair: the peaks, then the valleys inside
    # the span of THAT profile's peaks, then the two means) and ONE transfer of 16 doubles per profile
    _, _, _, _, means = ops.peak_valley_regions(
        p, [dict(threshold=0.3, peak_separation=sp, max_number=npk, search_region=(st, en)) for st, en, npk, _, sp, _ in regions],
        [dict(threshold=0.3, peak_separation=sp, max_number=nval) for _, _, _, nval, sp, _ in regions])
    return m, len(regions), ops.HostCopy(means.reshape(m, 2 * len(regions)))


def _ctp528_mtf_finish(pending):
    m, nr, copy = pending
    mh = copy.numpy().reshape(m, nr, 2)
    # a region counts while every region before it held its number of peaks (the reference's `break`): the peak mean is NaN
    # exactly where the count was wrong
    alive = np.logical_and.accumulate(~np.isnan(mh[:, :, 0]), axis=1)
    maxs = np.where(alive, mh[:, :, 0], np.nan)
    mins = np.where(alive, mh[:, :, 1], np.nan)
    nreg = alive.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mtf = (maxs - mins) / (maxs + mins)                             # michelson: (max - min) / (max + min)
        rmtf = mtf / mtf[:, :1]
    return dict(rmtf=rmtf, nregions=nreg, maxs=maxs, mins=mins)


def _device_centres_disagree(aux: np.ndarray, idx: np.ndarray, spv: int, fzx: np.ndarray, fzy: np.ndarray, h: int, w: int,
                             mm_per_pixel: float, kw: dict, rerun: bool):
    """Pass one sampled the profiles ``idx`` about the DEVICE's centres; the reference's are ``np.poly1d(np.polyfit(...))``.
    -> None when every profile provably holds the samples the exact centre selects (|difference| + a few roundings below the
    profile's decision margin), else the positions in ``idx`` to sample again (all of them when the ROI table itself changed
    on the host)."""
    m = len(idx)
    nflag = len(aux) - 3 * m
    cen, margin, flag = aux[:2 * m].reshape(m, 2), aux[2 * m:3 * m], aux[3 * m:3 * m + nflag]
    v, z = idx // spv, idx % spv
    cx = fzx[v, 0] * z + fzx[v, 1]                                             # np.poly1d(fit)(z), ct.py:434-439
    cy = fzy[v, 0] * z + fzy[v, 1]
    radius = kw.get("radius2linepairs_mm", 47) * kw.get("scaling_factor", 1.0) / mm_per_pixel
    if (w < radius + cx).any() or (h < radius + cy).any():                     # CircleProfile._ensure_array_size
        raise ValueError("Array size not large enough to compute profile")
    if rerun or flag.any():
        return np.arange(m)
    # a tap's coordinate is fl(fl(cos * r) + centre) and the decision reads fl(coordinate + 0.5): the product is the same
    # number on both sides, each of the two sums rounds by at most half an ulp of a coordinate (< max(h, w)) -- four
    # roundings in all between the two evaluations, covered sixteen times over by the slack
    slack = 16 * np.finfo(np.float64).eps * max(h, w)
    with np.errstate(invalid="ignore"):
        ok = np.maximum(np.abs(cen[:, 0] - cx), np.abs(cen[:, 1] - cy)) + slack < margin
    return None if ok.all() else np.flatnonzero(~ok)


def ctp528_batch(volume: torch.Tensor, mm_per_pixel: float, fit_zx=None, fit_zy=None, slices=None, roll_deg: float = 0.0,
                 chunk_volumes: int | None = None, **kw):
    """Config #5's per-slice record for resident CatPhan volumes: one volume [S, H, W] or several
    [V, S, H, W] in ONE batch (the phantom ROI of every slice, then per volume the reference's axis fits, then the circle
    profile and relative MTF of every requested slice; volumes never mix: the +-3-slice window and the fits stay inside a
    volume).  -> dict(center float64 [M, 2] = (x, y) fitted phantom centre, profiles float64 [M, L] (device),
    rmtf float64 [M, 8], nregions, maxs, mins, slices (indices into the flattened stack), roi = the per-slice phantom
    ROI table, fit_zx / fit_zy [V, 2])."""
    x = volume if volume.dim() == 4 else volume[None]
    nv, spv = x.shape[0], x.shape[1]
    hh, ww = x.shape[2], x.shape[3]
    # ``chunk_volumes`` volumes at a time (default: all at once): every chunk's localisation is queued before the host waits
    # for the first ROI table, each transfer is an event the host waits for only when it needs those numbers, so the host's
    # share of a chunk (two polynomial fits per volume, queueing launches) overlaps the device's work on the next.  Measured
    # on 25 volumes (scripts/run_ct_pass.py): one chunk 4.88 ms, two 4.95 ms, four 6.2 ms -- the per-slice workgroups of
    # pl_edge_regions fill the chip in whole rounds of 512 slices, which smaller launches waste; chunks are for bounding the
    # 1 MiB-per-slice edge plane, not for speed.
    chunk = int(chunk_volumes) if chunk_volumes else nv
    bounds = [(a, min(a + chunk, nv)) for a in range(0, nv, chunk)]
    given = fit_zx is not None and fit_zy is not None
    sl = None if slices is None else np.asarray(slices, dtype=np.int64)
    flats = [x[a:b].reshape((b - a) * spv, hh, ww) for a, b in bounds]
    gz = (np.atleast_2d(np.asarray(fit_zx, dtype=np.float64)), np.atleast_2d(np.asarray(fit_zy, dtype=np.float64))) if given else None
    # Pass one, queued without a single wait: the localisation of every chunk, and -- where the ROI table stays on the device
    # (16-bit slices) -- the placement fit (pl_phantom_axis_fit), the circle profiles about ITS centres and the MTF searches.
    # Round 5 stopped the device here for the ROI table, fitted on the host and only then queued the second half: 0.8 ms of
    # a 4.85 ms pass with nothing running.
    queued = []
    for (a, b), f in zip(bounds, flats):
        if sl is None:
            pos, local = None, None
        else:
            pos = np.flatnonzero((sl >= a * spv) & (sl < b * spv))
            local = sl[pos] - a * spv
        q = dict(a=a, b=b, f=f, pos=pos, local=local, roi=None if given else _phantom_roi_launch(f, mm_per_pixel), dev=None)
        table = getattr(q["roi"][1], "device_table", None) if q["roi"] is not None else None
        if table is not None and (local is None or len(local)):
            _, cen, flag = ops.phantom_axis_fit(table, b - a)
            prof, idx, margin = ctp528_profiles_batch(f, mm_per_pixel, None, None, slices=local, roll_deg=roll_deg,
                                                      slices_per_volume=spv, device_centers=cen, **kw)
            sel = cen if local is None else cen[_device_index(idx, cen.device)]
            aux = ops.HostCopy(torch.cat([sel.reshape(-1), margin, flag.to(torch.float64)]))
            q["dev"] = (prof, idx, _ctp528_mtf_launch(prof), aux)
        queued.append(q)
    second, where, per_chunk = [], [], []
    for q in queued:
        a, b, f, local = q["a"], q["b"], q["f"], q["local"]
        if given:
            roi, fzx, fzy = None, gz[0][a:b], gz[1][a:b]
        else:
            roi = _phantom_roi_finish(q["roi"])
            fzx, fzy = find_phantom_axes_batch(roi, b - a)                 # the reference's np.polyfit, bit for bit
        # every chunk keeps its ROI table and fits -- also one that holds none of the requested slices -- so that the
        # [V, 2] fit tables are indexed by the GLOBAL volume number below
        per_chunk.append((roi, np.atleast_2d(fzx), np.atleast_2d(fzy)))
        if q["pos"] is not None:
            where.append(q["pos"])
        if local is not None and len(local) == 0:
            continue
        if q["dev"] is not None:
            prof, idx, mtf_pending, aux = q["dev"]
            redo = _device_centres_disagree(aux.numpy(), idx, spv, np.atleast_2d(fzx), np.atleast_2d(fzy), hh, ww, mm_per_pixel,
                                            kw, rerun=bool((q["roi"][1].numpy()[:, 0] == 5).any()))
            if redo is None:
                second.append((prof, idx + a * spv, mtf_pending))
                continue
            if len(redo) < len(idx):                                       # (never observed: a tap within 1e-9 of a decision)
                fix, _ = ctp528_profiles_batch(f, mm_per_pixel, fzx, fzy, slices=idx[redo], roll_deg=roll_deg,
                                               slices_per_volume=spv, **kw)
                prof = prof.clone()
                prof[_device_index(redo, prof.device)] = fix
                second.append((prof, idx + a * spv, _ctp528_mtf_launch(prof)))
                continue
        prof, idx = ctp528_profiles_batch(f, mm_per_pixel, fzx, fzy, slices=local, roll_deg=roll_deg, slices_per_volume=spv, **kw)
        second.append((prof, idx + a * spv, _ctp528_mtf_launch(prof)))
    parts = [(prof, idx, _ctp528_mtf_finish(pend)) for prof, idx, pend in second]
    if not parts:
        raise ValueError("no slices selected")
    idx = np.concatenate([p[1] for p in parts])
    if given:
        fzx, fzy = gz
    else:
        fzx, fzy = np.concatenate([c[1] for c in per_chunk]), np.concatenate([c[2] for c in per_chunk])
    v, z = idx // spv, idx % spv
    out = {k: np.concatenate([p[2][k] for p in parts]) for k in ("rmtf", "nregions", "maxs", "mins")}
    prof = parts[0][0] if len(parts) == 1 else torch.cat([p[0] for p in parts])
    if sl is not None:                                     # back into the order the caller listed the slices in
        back = np.argsort(np.concatenate(where), kind="stable")
        if not np.array_equal(back, np.arange(len(back))):
            idx, v, z = idx[back], v[back], z[back]
            out = {k: a[back] for k, a in out.items()}
            prof = prof[torch.from_numpy(back).to(prof.device)]
    roi = None if given else np.concatenate([c[0] for c in per_chunk])
    out.update(center=np.stack([fzx[v, 0] * z + fzx[v, 1], fzy[v, 0] * z + fzy[v, 1]], axis=1), profiles=prof, slices=idx,
               roi=roi, fit_zx=fzx if volume.dim() == 4 else fzx[0], fit_zy=fzy if volume.dim() == 4 else fzy[0])
    return out
