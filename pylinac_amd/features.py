"""Batched BB / disk finder.

Mirrors ``pylinac.metrics.utils.find_features`` (pylinac/metrics/utils.py:66-190) with the detection
conditions of ``SizedDiskLocator`` (pylinac/metrics/image.py:529-535: is_right_size_bb, is_round,
is_right_circumference, is_symmetric, is_solid) for a batch of equally-sized float64 samples resident on
the GPU, and ``SizedDiskRegion.calculate``'s window + invert (metrics/image.py:564-612) for WL frames.

The 50-step threshold sweep runs level by level for the whole batch; a window stops taking part once it
has ``max_number`` features (the reference's ``while ... len(total_features) < max_number``).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import check


def stretch_device(frames: torch.Tensor) -> torch.Tensor:
    """``stretch(sample, 0, 1)`` = ground(normalize(ground(a)) * (1 - 0), value=0) (array_utils.py:168)."""
    g = ops.ground(frames)
    return ops.ground(ops.scale(ops.normalize(g), 1.0), value=0.0)


def sweep_cutoffs() -> list[float]:
    """cutoff = imin + step; while cutoff <= imax: ...; cutoff += step  with imin = 0, imax = 1 after
    stretch (metrics/utils.py:121-128, 180)."""
    imin, imax = 0.0, 1.0
    step = (imax - imin) / 50
    out, cutoff = [], imin + step
    while cutoff <= imax:
        out.append(cutoff)
        cutoff += step
    return out


SWEEP_MAX_SIDE = 160      # pl_features_sweep keeps a window of at most 160 x 160 samples in LDS


def find_features_batch(samples: torch.Tensor, dpmm: float, radius_mm: float, radius_tolerance_mm: float,
                        max_number: int = 1, min_separation_mm: float = 5, max_labels: int = 4096,
                        poll_every: int = 8, level_by_level: bool = False, defer: bool = False):
    """-> dict(xy float64 [N,8,2] (x, y) window coordinates, count int32 [N], level int32 [N],
    status int32 [N]).  ``count < min_number`` is the reference's ValueError("Couldn't find the minimum
    number of disks"); the batch reports it per window instead of raising.

    Windows up to 160 x 160 run the whole sweep in one launch (``pl_features_sweep``: one workgroup per window, the
    window resident in LDS); larger windows, and windows the sweep kernel hands back (status 3 / 5: a candidate or a
    level too large for its tables), take the level-by-level path (``level_by_level=True`` forces it).  ``defer=True``
    leaves windows with status 3 / 5 to the caller (no host synchronisation here)."""
    s = ops._frames(samples)
    if s.dtype != torch.float64:
        raise TypeError("find_features_batch needs float64 samples")
    n, h, w = s.shape
    dev = s.device
    s = stretch_device(s)
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    if not level_by_level and h <= SWEEP_MAX_SIDE and w <= SWEEP_MAX_SIDE and n > 0:
        cuts = np.ascontiguousarray(sweep_cutoffs(), dtype=np.float64)
        count = torch.empty(n, dtype=torch.int32, device=dev)
        level = torch.empty(n, dtype=torch.int32, device=dev)
        status = torch.empty(n, dtype=torch.int32, device=dev)
        xy = torch.empty((n, 8, 2), dtype=torch.float64, device=dev)
        check(lib.pl_features_sweep(s.data_ptr(), n, h, w, float(dpmm), float(radius_mm), float(radius_tolerance_mm),
                                    float(min_separation_mm * dpmm), int(max_number), cuts.ctypes.data, len(cuts),
                                    count.data_ptr(), xy.data_ptr(), level.data_ptr(), status.data_ptr(), st),
              "pl_features_sweep")
        if defer:
            return dict(xy=xy, count=count, level=level, status=status)
        redo = torch.nonzero((status == 3) | (status == 5)).flatten()
        if redo.numel():                                        # tables too small for these windows: the general path
            sub = _find_features_levels(s[redo].contiguous(), dpmm, radius_mm, radius_tolerance_mm, max_number,
                                        min_separation_mm, max_labels, poll_every)
            xy[redo], count[redo], level[redo], status[redo] = sub["xy"], sub["count"], sub["level"], sub["status"]
        return dict(xy=xy, count=count, level=level, status=status)
    return _find_features_levels(s, dpmm, radius_mm, radius_tolerance_mm, max_number, min_separation_mm, max_labels,
                                 poll_every)


def _find_features_levels(s: torch.Tensor, dpmm: float, radius_mm: float, radius_tolerance_mm: float, max_number: int,
                          min_separation_mm: float, max_labels: int, poll_every: int):
    """the level-by-level sweep over stretched samples ``s`` (compare -> label -> region table -> pl_features_level)"""
    n, h, w = s.shape
    dev = s.device
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    done = torch.zeros(n, dtype=torch.int32, device=dev)
    count = torch.zeros(n, dtype=torch.int32, device=dev)
    level = torch.full((n,), -1, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    xy = torch.zeros((n, 8, 2), dtype=torch.float64, device=dev)
    for lvl, cutoff in enumerate(sweep_cutoffs()):
        bw = ops.compare(s, cutoff, ">")
        labels, num = ops.label(bw, 4)
        stats, _ = ops.region_stats(labels, None, max_labels)
        check(lib.pl_features_level(s.data_ptr(), labels.data_ptr(), num.data_ptr(), stats.data_ptr(), max_labels, n, h,
                                    w, float(dpmm), float(radius_mm), float(radius_tolerance_mm),
                                    float(min_separation_mm * dpmm), int(max_number), lvl, done.data_ptr(),
                                    count.data_ptr(), xy.data_ptr(), level.data_ptr(),
                                    status.data_ptr(), st), "pl_features_level")
        if poll_every and (lvl + 1) % poll_every == 0 and bool(done.all()):
            break
    return dict(xy=xy, count=count, level=level, status=status)


def bb_centroids_batch(frames: torch.Tensor, dpmm: float, bb_diameter_mm: float, low_density: bool = False,
                       bb_tolerance_mm: float | None = None, vmin: torch.Tensor | None = None,
                       vmax: torch.Tensor | None = None, defer: bool = False, shift: bool = True):
    """``WLBaseImage.find_bb_centroids`` (pylinac/winston_lutz.py:788-806) for uint16 frames:
    SizedDiskLocator.from_center_physical(expected (0, 0) mm, window (40 + d) mm, radius d/2,
    invert = not low_density) on the ground()/normalize()d frame.  Returns the find_features result
    with ``xy`` shifted to frame coordinates (``shift=False``: left in window coordinates, the offsets in ``window``)."""
    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        raise TypeError("bb_centroids_batch needs uint16 frames")
    n, h, w = x.shape
    if bb_tolerance_mm is None:                         # _calculate_bb_tolerance, winston_lutz.py:1062-1067
        bb_tolerance_mm = float(np.interp(bb_diameter_mm, (1.5, 30), (2, 4)))
    win = (40 + bb_diameter_mm) * dpmm                  # search window in pixels
    ex, ey = w / 2, h / 2                               # expected position (0, 0) mm from the centre
    left = max(math.floor(ex - win / 2), 0)
    right = math.ceil(ex + win / 2)
    top = max(math.floor(ey - win / 2), 0)
    bottom = math.ceil(ey + win / 2)
    if vmin is None or vmax is None:
        vmin, vmax = ops.minmax(x)                      # frame-level ground()/normalize()
    vmin, vmax = vmin.to(torch.float64).contiguous(), vmax.to(torch.float64).contiguous()
    bottom, right = min(bottom, h), min(right, w)       # numpy slicing clips at the frame's edge
    wh, ww = bottom - top, right - left
    if wh <= SWEEP_MAX_SIDE and ww <= SWEEP_MAX_SIDE and n > 0:
        # the window straight from the uint16 frames: pl_features_sweep_u16 evaluates ground / normalize / invert / stretch
        # per pixel inside the sweep's workgroup (bit-identical to the separate kernels below; no float64 window in HBM)
        dev = x.device
        cuts = np.ascontiguousarray(sweep_cutoffs(), dtype=np.float64)
        count = torch.empty(n, dtype=torch.int32, device=dev)
        level = torch.empty(n, dtype=torch.int32, device=dev)
        status = torch.empty(n, dtype=torch.int32, device=dev)
        xy = torch.empty((n, 8, 2), dtype=torch.float64, device=dev)
        check(_lib.load().pl_features_sweep_u16(x.data_ptr(), n, h, w, top, left, wh, ww, vmin.data_ptr(), vmax.data_ptr(),
                                                0 if low_density else 1, float(dpmm), float(bb_diameter_mm / 2),
                                                float(bb_tolerance_mm), float(5 * dpmm), 1, cuts.ctypes.data, len(cuts),
                                                count.data_ptr(), xy.data_ptr(), level.data_ptr(), status.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "pl_features_sweep_u16")
        res = dict(xy=xy, count=count, level=level, status=status)
        if not defer:
            redo = torch.nonzero((status == 3) | (status == 5)).flatten()
            if redo.numel():                                    # tables too small for these windows: the general path
                sub = _bb_sample(x.view(torch.int16)[redo].view(torch.uint16), top, bottom, left, right, vmin[redo], vmax[redo],
                                 low_density)
                r2 = find_features_batch(sub, dpmm, bb_diameter_mm / 2, bb_tolerance_mm, level_by_level=True)
                for key in ("xy", "count", "level", "status"):
                    res[key][redo] = r2[key]
    else:
        sample = _bb_sample(x, top, bottom, left, right, vmin, vmax, low_density)
        res = find_features_batch(sample, dpmm, bb_diameter_mm / 2, bb_tolerance_mm, defer=defer)
    if shift:                                            # (scalar adds: no host-to-device copy, no synchronisation)
        res["xy"][..., 0] += float(left)
        res["xy"][..., 1] += float(top)
    res["window"] = (top, bottom, left, right)           # shift=False: ``xy`` stays in window coordinates, the caller adds these
    return res


def _bb_sample(x: torch.Tensor, top: int, bottom: int, left: int, right: int, vmin: torch.Tensor, vmax: torch.Tensor,
               low_density: bool) -> torch.Tensor:
    """the float64 sample of the BB window as separate kernels: crop, frame-level ground / normalize, invert"""
    crop = x.view(torch.int16)[:, top:bottom, left:right].contiguous().view(torch.uint16)
    q = ops.normalize(ops.ground(crop, mn=vmin), vmax - vmin)        # float64 (a - min) / (max - min)
    return ops.invert(q) if not low_density else q


def field_cutoffs(imin: float, imax: float) -> list[float]:
    """The threshold ladder of GlobalSizedFieldLocator.calculate (pylinac/metrics/image.py:836-842, 889):
    50 steps of the spread, starting at 10 % height, accumulated by repeated addition like the reference."""
    step = (imax - imin) / 50
    cutoff = imin + step * 5
    out = []
    while cutoff <= imax and len(out) < 64:
        out.append(cutoff)
        cutoff += step
    return out


def find_fields_batch(frames: torch.Tensor, dpmm: float, field_width_mm: float, field_height_mm: float,
                      field_tolerance_mm: float, max_number: int | None = None, is_from_physical: bool = True,
                      max_labels: int = 4096):
    """Batched ``GlobalSizedFieldLocator.calculate`` (pylinac/metrics/image.py:817-897; driven for multi-target WL
    at pylinac/winston_lutz.py:2734-2766) -> dict(xy float64 [N,8,2] (x, y), count, level, status int32 [N]).

    Per level: ``frame > cutoff`` -> 8-connected labels -> region table -> ``pl_fields_level`` (border band,
    perimeter / filled-area predicates, unweighted centroid, the level's own de-duplication radius).  A frame stops
    once ``max_number`` fields are found; ``count < min_number`` is the reference's ValueError, reported per frame.
    """
    f = ops._frames(frames)
    n, h, w = f.shape
    dev = f.device
    if not is_from_physical:                       # image.py:829-832: sizes given in pixels
        field_width_mm, field_height_mm = field_width_mm / dpmm, field_height_mm / dpmm
        field_tolerance_mm = field_tolerance_mm / dpmm
    max_number = int(max_number or 8)
    if max_number > 8:
        raise ValueError("at most 8 fields per frame are reported")
    mn, mx = ops.minmax(f)
    lad = [field_cutoffs(a, b) for a, b in zip(mn.cpu().tolist(), mx.cpu().tolist())]
    levels = max((len(x) for x in lad), default=0)
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    done = torch.zeros(n, dtype=torch.int32, device=dev)
    count = torch.zeros(n, dtype=torch.int32, device=dev)
    level = torch.full((n,), -1, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    xy = torch.zeros((n, 8, 2), dtype=torch.float64, device=dev)
    for lvl in range(levels):
        # a frame whose ladder ended gets +inf: an empty mask, like the reference's finished while-loop
        cut = torch.tensor([x[lvl] if lvl < len(x) else float("inf") for x in lad], dtype=torch.float64, device=dev)
        bw = ops.compare(f, cut, ">")
        labels, num = ops.label(bw, 8)
        stats, _ = ops.region_stats(labels, None, max_labels)
        check(lib.pl_fields_level(labels.data_ptr(), num.data_ptr(), stats.data_ptr(), max_labels, n, h, w,
                                  float(dpmm), float(field_width_mm), float(field_height_mm),
                                  float(field_tolerance_mm), 3, max_number, lvl, done.data_ptr(), count.data_ptr(),
                                  xy.data_ptr(), level.data_ptr(), status.data_ptr(), st), "pl_fields_level")
        if (lvl + 1) % 8 == 0 and bool(done.all()):
            break
    return dict(xy=xy, count=count, level=level, status=status)
