"""Batched per-image part of Winston-Lutz: the field CAX.

Mirrors, for every frame of a device-resident uint16/int16 batch, the sequence of
``WLBaseImage.analyze`` -> ``find_field_centroids`` (pylinac/winston_lutz.py:711-712, 764-780):

    self.ground(); self.normalize()                                    # -> float64 frame n
    min, max = np.percentile(self.array, [5, 99.9])
    threshold_img = self.as_binary((max - min) / 2 + min)
    filled_img = ndimage.binary_fill_holes(threshold_img)
    coords = ndimage.center_of_mass(filled_img);  Point(x=coords[-1], y=coords[0])

The float64 frame is never materialised: ground()/normalize() are monotone, so the percentiles'
order statistics are taken from the exact 16-bit histogram and pushed through the same float64
operations; the binary image is produced directly from the integer frame by
``((a - min) / max') >= t`` in float64 (``pl_scaled_binary``).

``analyze_batch`` composes the whole per-image sequence of ``WLBaseImage.analyze`` (winston_lutz.py:709-725):
inversion check -> ``_clean_edges`` -> ground / normalize -> field CAX -> BB sweep, one ``(field_x, field_y, bb_x,
bb_y)`` record per frame.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops


_FIELD_Q = (5.0, 99.9)          # find_field_centroids, winston_lutz.py:775
_INVERSION_Q = (0.01, 50.0, 99.99)   # analyze, winston_lutz.py:709
_EDGE_Q = (5.0, 99.5)            # _clean_edges, winston_lutz.py:1117


class _FrameStats:
    """Every order statistic the per-image sequence asks for, from ONE exact histogram per frame and ONE selection
    launch: min, max and the two neighbours of each percentile's virtual index; ``np.percentile``'s interpolation
    (``_lerp``) is evaluated on the host table."""

    def __init__(self, x: torch.Tensor, hist: torch.Tensor | None = None, qs=_FIELD_Q + _INVERSION_Q + _EDGE_Q):
        self.cnt = x[0].numel()
        self.qs = tuple(qs)
        _, lo, hi, frac = ops._percentile_plan(self.cnt, list(self.qs))
        self.frac = frac
        ranks = np.concatenate([[0, self.cnt - 1], lo, hi])
        hist = ops.histogram16(x) if hist is None else hist
        st = ops.order_stats(x, ranks, hist=hist).cpu().numpy().astype(np.float64)
        # contiguous copies: these columns travel to the device through raw pointers
        self.vmin, self.vmax = np.ascontiguousarray(st[:, 0]), np.ascontiguousarray(st[:, 1])
        k = len(self.qs)
        self.lo, self.hi = np.ascontiguousarray(st[:, 2:2 + k]), np.ascontiguousarray(st[:, 2 + k:2 + 2 * k])

    def percentiles(self, qs, transform=None) -> np.ndarray:
        """np.percentile(f(frame), qs) for a monotone elementwise ``transform`` f applied to the order statistics
        (identity when None) -> float64 [N, len(qs)]"""
        idx = [self.qs.index(q) for q in qs]
        a, b = self.lo[:, idx], self.hi[:, idx]
        if transform is not None:
            a, b = transform(a), transform(b)
        t = self.frac[idx][None, :]
        d = b - a
        return np.where(t >= 0.5, b - d * (1 - t), a + d * t)


def _field_threshold(stats: _FrameStats):
    """ground() / normalize() then ``(p99.9 - p5) / 2 + p5`` of the float64 frame (winston_lutz.py:711-712, 775-776):
    the order statistics pushed through the same float64 operations -> (vmin, gmax, thr) float64 [N] on the host"""
    vmin, gmax = stats.vmin, stats.vmax - stats.vmin            # max of the grounded frame
    p = stats.percentiles(_FIELD_Q, transform=lambda v: (v - vmin[:, None]) / gmax[:, None])
    return vmin, gmax, (p[:, 1] - p[:, 0]) / 2 + p[:, 0]


def field_centroids_batch(frames: torch.Tensor, stats: _FrameStats | None = None) -> torch.Tensor:
    """-> float64 [N, 3] = (x, y, filled_pixel_count) of the field centroid of every frame.  ``stats``: the frames'
    order statistics when the caller already has them."""
    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        # int16: the reference's ground() (`array - array.min()`, array_utils.py:102) wraps around in
        # int16 for any frame whose range exceeds 32767, i.e. its own result is an overflow artefact
        raise TypeError("field_centroids_batch needs uint16 frames")
    stats = _FrameStats(x, qs=_FIELD_Q) if stats is None else stats
    vmin, gmax, thr = (torch.from_numpy(np.ascontiguousarray(v)).to(x.device) for v in _field_threshold(stats))
    cen = ops.field_cax(x, vmin, gmax, thr)                    # threshold -> fill holes -> centre of mass: row, col, count
    return torch.stack([cen[:, 1], cen[:, 0], cen[:, 2]], dim=1)


def analyze_batch(frames: torch.Tensor, dpmm: float, bb_diameter_mm: float = 5.0, low_density: bool = False,
                  clean_edges: bool = True, tile_maxima: bool = True):
    """The per-image part of ``WLBaseImage.analyze`` (pylinac/winston_lutz.py:709-725) for a batch of uint16 frames
    resident on the GPU (float64 and int16 frames: see the first lines of the body):

        check_inversion_by_histogram((0.01, 50, 99.99))   image.py:899-926
        _clean_edges()                                     winston_lutz.py:1109-1133
        ground(); normalize()                              winston_lutz.py:711-712
        find_field_centroids(is_open_field=False)          winston_lutz.py:764-780
        find_bb_centroids(bb_diameter_mm, low_density)     winston_lutz.py:788-806

    -> dict(record float64 [N, 4] = (field_x, field_y, bb_x, bb_y) in the coordinates of the (possibly edge-cleaned)
    frame, like the reference's points; status int32 [N]: 0 ok, 1 = no BB found (the reference raises ValueError);
    inverted bool [N]; crop int32 [N] = pixels ``_clean_edges`` removed from every side).

    Round 3: ONE host synchronisation per batch.  Every frame first runs the sequence as if it were neither inverted nor
    in need of edge cleaning -- one exact histogram + one selection launch serve every percentile, ``pl_wl_decisions`` takes
    the three scalar decisions on the device, the field CAX and the BB sweep read their thresholds from device arrays --
    and a single table (records, flags, kernel status words) comes back.  The frames whose flags say otherwise (inverted
    polarity, dirty edges, a field or BB window too large for a kernel's LDS tables) are then finished by the
    host-driven sequence ``_analyze_batch_host`` (the round-2 path), exactly as before."""
    from . import features

    x = ops._frames(frames)
    if x.dtype in (torch.float64, torch.int16):
        # what the reference's loader hands over once RescaleSlope / RescaleIntercept exist (float64: pydicom's
        # apply_rescale, pylinac/core/image.py:375-389) or for a signed panel: the same sequence on the float
        # kernels (``_analyze_batch_general``) -- several passes per frame instead of one and a bit, but resident and exact
        return _analyze_batch_general(x, dpmm, bb_diameter_mm, low_density, clean_edges)
    if x.dtype != torch.uint16:
        raise TypeError("analyze_batch takes uint16, int16 or float64 frames (float32 never leaves the reference's loader: "
                        "apply_rescale widens to float64)")
    n, h, w = x.shape
    if n == 0:
        return dict(record=np.zeros((0, 4)), status=np.zeros(0, np.int32), inverted=np.zeros(0, bool), crop=np.zeros(0, np.int32))
    cnt = h * w
    qs = _FIELD_Q + _INVERSION_Q + _EDGE_Q
    _, lo, hi, frac = ops._percentile_plan(cnt, list(qs))
    # the histogram pass also notes the largest value of every 512-pixel tile: the field CAX then reads only the tiles that
    # can hold a pixel above the field threshold instead of the whole batch a second time (``tile_maxima=False``: the full
    # pass; same results -- an A/B knob)
    # ... and the min / max of the 2-pixel edge strips (``_clean_edges``' test) come out of the same launch
    # and so do the order statistics of every percentile the sequence asks for (selected from the histogram while it is in LDS)
    ranks = np.concatenate([[0, cnt - 1], lo, hi])
    if tile_maxima:
        _, tmax, emin, emax, st = ops.histogram16(x, tiles=True, edge_window=2, ranks=ranks)     # st: int32 [N, 16], on the device
    else:
        hist, tmax = ops.histogram16(x), None
        emin, emax = ops.edge_minmax(x, 2)
        st = ops.order_stats(x, ranks, hist=hist)
        del hist
    dec = ops.wl_decisions(st, emin, emax, frac)
    cen, cax_status = ops.field_cax(x, dec["vmin"], dec["gmax"], dec["thr"], defer=True, tile_max=tmax)   # (row, col, count)
    bb = features.bb_centroids_batch(x, dpmm, bb_diameter_mm, low_density=low_density, vmin=dec["vmin"], vmax=dec["vmax"],
                                     defer=True, shift=False)
    top, _, left, _ = bb["window"]
    # the record table in ONE launch (field x / y, BB x / y moved from window to frame coordinates, count, the status words,
    # the two decision flags) and the one synchronisation of the pass
    table = ops.pack_columns([(cen, 1, 0.0), (cen, 0, 0.0), (bb["xy"], 0, float(left)), (bb["xy"], 1, float(top)), bb["count"],
                              bb["status"], cax_status, dec["inverted"], dec["noisy"]], n).cpu().numpy()
    cntb = table[:, 4]
    record = np.concatenate([table[:, :2], np.where(cntb[:, None] > 0, table[:, 2:4], np.nan)], axis=1)
    status = (cntb == 0).astype(np.int32)
    inverted = table[:, 7] != 0
    crop = np.zeros(n, dtype=np.int32)
    redo = inverted | (table[:, 6] != 0) | (table[:, 5] == 3) | (table[:, 5] == 5)
    if clean_edges:
        redo |= table[:, 8] != 0
    if redo.any():
        idx = np.nonzero(redo)[0]
        sub = x.view(torch.int16)[torch.from_numpy(idx).to(x.device)].view(torch.uint16)
        r = _analyze_batch_host(sub, dpmm, bb_diameter_mm, low_density, clean_edges)
        record[idx], status[idx], inverted[idx], crop[idx] = r["record"], r["status"], r["inverted"], r["crop"]
    return dict(record=record, status=status, inverted=inverted, crop=crop)


def _analyze_batch_host(frames: torch.Tensor, dpmm: float, bb_diameter_mm: float = 5.0, low_density: bool = False,
                        clean_edges: bool = True):
    """The host-driven form of ``analyze_batch`` (round 2): every decision is taken on the host from the order-statistics
    table, which costs a handful of synchronisations per call.  ``analyze_batch`` hands it the frames that leave the common
    path (inverted polarity, dirty edges, oversized windows).  Same sequence:

        check_inversion_by_histogram((0.01, 50, 99.99))   image.py:899-926
        _clean_edges()                                     winston_lutz.py:1109-1133
        ground(); normalize()                              winston_lutz.py:711-712
        find_field_centroids(is_open_field=False)          winston_lutz.py:764-780
        find_bb_centroids(bb_diameter_mm, low_density)     winston_lutz.py:788-806

    -> dict(record float64 [N, 4] = (field_x, field_y, bb_x, bb_y) in the coordinates of the (possibly edge-cleaned)
    frame, like the reference's points; status int32 [N]: 0 ok, 1 = no BB found (the reference raises ValueError);
    inverted bool [N]; crop int32 [N] = pixels ``_clean_edges`` removed from every side).

    One exact histogram and one selection launch per batch serve every percentile the sequence asks for; the decisions
    (three comparisons per frame) are taken on the host from that table, inversion is applied on the device to the
    frames that need it.  A frame whose edges need cleaning changes shape: it is finished on its own (same kernels,
    batch of one)."""
    from . import decisions, features

    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        raise TypeError("analyze_batch needs uint16 frames")
    n, h, w = x.shape
    dev = x.device
    stats = _FrameStats(x)
    # ---- inversion (|p50 - p0.01| > |p50 - p99.99|): invert -a + max + min, in the frame's dtype
    p = stats.percentiles(_INVERSION_Q)
    inverted = np.abs(p[:, 1] - p[:, 0]) > np.abs(p[:, 1] - p[:, 2])
    if inverted.any():
        idx = torch.from_numpy(np.nonzero(inverted)[0]).to(dev)
        x = x.clone()
        xi = x.view(torch.int16)                       # torch has no indexed copies for uint16: same bits as int16
        xi[idx] = ops.invert(xi[idx].view(torch.uint16)).view(torch.int16)
        stats = _FrameStats(x)                         # the inverted frames' order statistics (a rare branch)
    # ---- edge cleaning decision on the whole batch; frames that need cropping leave the batch
    crop = np.zeros(n, dtype=np.int32)
    record = np.full((n, 4), np.nan, dtype=np.float64)
    status = np.zeros(n, dtype=np.int32)
    keep = np.ones(n, dtype=bool)
    if clean_edges:
        pe = stats.percentiles(_EDGE_Q)
        emin, emax = (t.cpu().numpy().astype(np.float64) for t in ops.edge_minmax(x, 2))
        rng = pe[:, 1] - pe[:, 0]
        noisy = (emin < pe[:, 0] - rng / 10) | (emax > pe[:, 1] + rng / 10)
        for i in np.nonzero(noisy)[0]:
            cleaned = decisions.clean_edges(x[i])                              # the reference's loop, one frame
            crop[i] = (h - cleaned.shape[0]) // 2
            one = _analyze_batch_host(cleaned[None], dpmm, bb_diameter_mm, low_density, clean_edges=False)
            record[i], status[i] = one["record"][0], one["status"][0]
            keep[i] = False
    if keep.any():
        if keep.all():
            sel, sst = x, stats
        else:
            sel = x.view(torch.int16)[torch.from_numpy(np.nonzero(keep)[0]).to(dev)].view(torch.uint16)
            sst = _FrameStats(sel)
        fld = field_centroids_batch(sel, stats=sst).cpu().numpy()
        bb = features.bb_centroids_batch(sel, dpmm, bb_diameter_mm, low_density=low_density,
                                         vmin=torch.from_numpy(sst.vmin).to(dev), vmax=torch.from_numpy(sst.vmax).to(dev))
        bxy = bb["xy"][:, 0, :].cpu().numpy()
        cnt = bb["count"].cpu().numpy()
        rec = np.concatenate([fld[:, :2], np.where(cnt[:, None] > 0, bxy, np.nan)], axis=1)
        record[keep] = rec
        status[keep] = (cnt == 0).astype(np.int32)
    return dict(record=record, status=status, inverted=inverted, crop=crop)


def _percentiles_f64(x: torch.Tensor, qs) -> np.ndarray:
    """``np.percentile(frame, qs)`` per float64 frame -> float64 [N, len(qs)] on the host (exact order statistics by key
    bisection, ``pl_order_stats_f64``; numpy's ``_lerp``)."""
    from .canny import _percentile_f64

    return np.stack([_percentile_f64(x, float(q)).cpu().numpy() for q in qs], axis=1)


def _edge_strip_extrema(x: torch.Tensor, ws: int = 2):
    """min / max over the four ``ws``-wide edge strips of every float64 frame (``_clean_edges``, winston_lutz.py:1121-1127)"""
    from .roi import rectangle_stats_batch

    n, h, w = x.shape
    strips = np.array([[0, ws, 0, w], [0, h, 0, ws], [h - ws, h, 0, w], [0, h, w - ws, w]], dtype=np.float64)
    s = rectangle_stats_batch(x, strips)[0].cpu().numpy()          # [N, 4, (count, mean, std, min, max, ...)]
    return s[:, :, 3].min(axis=1), s[:, :, 4].max(axis=1)


def _analyze_batch_general(frames: torch.Tensor, dpmm: float, bb_diameter_mm: float = 5.0, low_density: bool = False,
                           clean_edges: bool = True, check_inversion: bool = True):
    """``analyze_batch`` for float64 frames (and int16 frames, widened exactly): every step of ``WLBaseImage.analyze``
    (winston_lutz.py:709-725) as the reference performs it on such an array -- ``np.percentile`` on the float values,
    ``-a + max + min``, ``a - min``, ``a / max``, ``as_binary``, ``binary_fill_holes`` + ``center_of_mass``,
    ``SizedDiskLocator`` on the float window -- with the general float kernels; the frames stay on the device.  Same result
    dictionary as ``analyze_batch``.

    int16: the reference's own ``ground()`` (``array - array.min()`` IN int16, array_utils.py:102) wraps around for a frame
    whose range exceeds 32767 and everything after it is an overflow artefact; such a frame raises ``ValueError`` here.  Below
    that the integer steps are exact in float64 (``invert`` wraps in int16 too, but its result lies inside [min, max])."""
    from . import features

    x = ops._frames(frames)
    n, h, w = x.shape
    if n == 0:
        return dict(record=np.zeros((0, 4)), status=np.zeros(0, np.int32), inverted=np.zeros(0, bool), crop=np.zeros(0, np.int32))
    if x.dtype == torch.int16:
        mn, mx = ops.minmax(x)
        if bool(((mx - mn) > 32767).any()):
            raise ValueError("int16 frame with a range beyond 32767: the reference's ground() overflows int16 there "
                             "(array_utils.py:102) and its result is an artefact; widen the frames first")
        x = ops.normalize(x, 1.0)                                   # exact conversion to float64
    dev = x.device
    # ---- check_inversion_by_histogram((0.01, 50, 99.99)), image.py:899-926
    inverted = np.zeros(n, dtype=bool)
    if check_inversion:
        p = _percentiles_f64(x, _INVERSION_Q)
        inverted = np.abs(p[:, 1] - p[:, 0]) > np.abs(p[:, 1] - p[:, 2])
    if inverted.any():
        idx = torch.from_numpy(np.nonzero(inverted)[0]).to(dev)
        x = x.clone()
        x[idx] = ops.invert(x[idx].contiguous())
    crop = np.zeros(n, dtype=np.int32)
    record = np.full((n, 4), np.nan, dtype=np.float64)
    status = np.zeros(n, dtype=np.int32)
    keep = np.ones(n, dtype=bool)
    # ---- _clean_edges, winston_lutz.py:1109-1133: frames that need cropping change shape and are finished on their own
    if clean_edges:
        pe = _percentiles_f64(x, _EDGE_Q)
        emin, emax = _edge_strip_extrema(x, 2)
        rng = pe[:, 1] - pe[:, 0]
        noisy = (emin < pe[:, 0] - rng / 10) | (emax > pe[:, 1] + rng / 10)
        for i in np.nonzero(noisy)[0]:
            f = x[i]
            safety_stop = min(f.shape) / 10
            while safety_stop > 0:
                f = f[2:-2, 2:-2].contiguous()                     # BaseImage.crop(window_size)
                safety_stop -= 1
                q = _percentiles_f64(f[None], _EDGE_Q)[0]
                lo_, hi_ = _edge_strip_extrema(f[None], 2)
                r_ = q[1] - q[0]
                if not (lo_[0] < q[0] - r_ / 10 or hi_[0] > q[1] + r_ / 10):
                    break
            crop[i] = (h - f.shape[0]) // 2
            # (the inversion test belongs to the uncropped frame and has been applied: winston_lutz.py:709-710 precede :711)
            one = _analyze_batch_general(f[None], dpmm, bb_diameter_mm, low_density, clean_edges=False, check_inversion=False)
            record[i], status[i] = one["record"][0], one["status"][0]
            keep[i] = False
    if keep.any():
        sel = x if keep.all() else x[torch.from_numpy(np.nonzero(keep)[0]).to(dev)].contiguous()
        nrm = ops.normalize(ops.ground(sel))                        # ground(); normalize(), winston_lutz.py:711-712
        # ---- find_field_centroids, winston_lutz.py:764-780
        fp = _percentiles_f64(nrm, _FIELD_Q)
        thr = torch.from_numpy(np.ascontiguousarray((fp[:, 1] - fp[:, 0]) / 2 + fp[:, 0])).to(dev)
        cen = ops.binary_centroid(ops.fill_holes(ops.as_binary(nrm, thr), 4)).cpu().numpy()      # (row, col, count)
        # ---- find_bb_centroids, winston_lutz.py:788-806 (SizedDiskLocator.from_center_physical about the image centre)
        tol = float(np.interp(bb_diameter_mm, (1.5, 30), (2, 4)))
        hh, ww = sel.shape[1], sel.shape[2]
        win = (40 + bb_diameter_mm) * dpmm
        ex, ey = ww / 2, hh / 2
        left, right = max(math.floor(ex - win / 2), 0), min(math.ceil(ex + win / 2), ww)
        top, bottom = max(math.floor(ey - win / 2), 0), min(math.ceil(ey + win / 2), hh)
        sample = nrm[:, top:bottom, left:right].contiguous()
        if not low_density:
            sample = ops.invert(sample)
        bb = features.find_features_batch(sample, dpmm, bb_diameter_mm / 2, tol)
        bxy = bb["xy"][:, 0, :].cpu().numpy() + np.array([left, top], dtype=np.float64)
        cnt = bb["count"].cpu().numpy()
        record[keep] = np.concatenate([cen[:, [1, 0]], np.where(cnt[:, None] > 0, bxy, np.nan)], axis=1)
        status[keep] = (cnt == 0).astype(np.int32)
    return dict(record=record, status=status, inverted=inverted, crop=crop)
