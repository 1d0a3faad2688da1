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

import numpy as np
import torch

from . import ops

CATPHAN_RADIUS_MM = 101  # pylinac/ct.py: CatPhanBase.catphan_radius_mm


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


def otsu_from_counts(counts: np.ndarray, centers: np.ndarray) -> float:
    """skimage 0.18.3 ``threshold_otsu`` on a ready histogram (filters/thresholding.py)."""
    counts = counts.astype(float)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * centers) / weight1
    mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    return centers[np.argmax(variance12)]


def get_regions_batch(slices: torch.Tensor, mm_per_pixel: float, fill_holes: bool = True,
                      clear_borders: bool = True, max_labels: int = 64):
    """-> dict(edges f64 [N,H,W], bw u8, labels i32, num i32 [N], stats f64 [N,max_labels,10],
    overflow i32 [N], otsu f64 [N])."""
    x = ops._frames(slices)
    n, h, w = x.shape
    dev = x.device
    edges = ops.gaussian_filter_mode(ops.scharr(x), 1, "nearest")
    cy, cx = h / 2 - 0.5, w / 2 - 0.5                                  # BaseImage.center, image.py:527-533
    disk = torch.from_numpy(disk_mask((cy, cx), 110 / mm_per_pixel, (h, w))).to(dev)
    # np.histogram(edges[disk], 256): the range is the min/max of the SELECTED pixels
    mn, mx = ops.minmax_masked(edges, disk)
    lo, hi = mn.cpu().numpy(), mx.cpu().numpy()
    e = np.stack([np.linspace(a, b, 257) for a, b in zip(lo, hi)])
    counts = ops.hist_uniform(edges, torch.from_numpy(e).to(dev), disk).cpu().numpy()
    otsu = np.array([a if a == b else otsu_from_counts(c, (ee[:-1] + ee[1:]) / 2.0)
                     for a, b, c, ee in zip(lo, hi, counts, e)])
    thr = torch.from_numpy(otsu * 0.8).to(dev)
    bw = ops.compare(edges, thr, ">")
    if clear_borders:
        bw = ops.clear_border(bw, min(int(max(h, w) / 100), 3))
    if fill_holes:
        bw = ops.fill_holes(bw, 4)
    labels, num = ops.label(bw, 8)
    stats, ovf = ops.region_stats(labels, edges, max_labels)
    return dict(edges=edges, bw=bw, labels=labels, num=num, stats=stats, overflow=ovf, otsu=otsu)


def phantom_roi_batch(slices: torch.Tensor, mm_per_pixel: float, catphan_radius_mm: float = CATPHAN_RADIUS_MM,
                      max_labels: int = 64) -> np.ndarray:
    """``Slice.phantom_roi`` for every slice -> float64 [N, 8]:
    status, label, filled_area, centroid_row, centroid_col, bbox(r0, c0, r1, c1)[first 3 shown]...
    columns: status (0 ok, 1 no edges, 2 no ROI, 3 wrong size, 4 label overflow), label,
    filled_area, centroid_r, centroid_c, bbox_r0, bbox_c0, bbox_r1 -- the reference raises ValueError
    for status 1-3; the batch reports per-slice codes instead."""
    x = ops._frames(slices)
    n = x.shape[0]
    catphan_size = np.pi * catphan_radius_mm**2 / mm_per_pixel**2        # ct.py:2581-2584
    raw_max = ops.minmax(ops.scharr(x))[1].cpu().numpy()                 # ct.py:392: np.max(edges) < 0.1
    reg = get_regions_batch(x, mm_per_pixel, fill_holes=True, clear_borders=True, max_labels=max_labels)
    stats = reg["stats"].cpu().numpy()
    num = reg["num"].cpu().numpy()
    ovf = reg["overflow"].cpu().numpy()
    out = np.full((n, 8), np.nan)
    for i in range(n):
        if raw_max[i] < 0.1:
            out[i, 0] = 1
            continue
        if num[i] < 1:
            out[i, 0] = 2
            continue
        if ovf[i]:
            out[i, 0] = 4
            continue
        t = stats[i, : num[i]]
        filled = t[:, 0]                                   # == filled_area after binary_fill_holes
        k = int(np.argsort(np.abs(filled - catphan_size), kind="stable")[0])
        if catphan_size * 1.3 < filled[k] or filled[k] < catphan_size / 1.3:
            out[i, 0] = 3
            continue
        out[i] = [0, k + 1, filled[k], t[k, 5] / t[k, 0], t[k, 6] / t[k, 0], t[k, 1], t[k, 2], t[k, 3]]
    return out
