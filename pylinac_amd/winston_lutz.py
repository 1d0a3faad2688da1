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

import numpy as np
import torch

from . import ops


def field_centroids_batch(frames: torch.Tensor, hist: torch.Tensor | None = None) -> torch.Tensor:
    """-> float64 [N, 3] = (x, y, filled_pixel_count) of the field centroid of every frame.  ``hist``: the frames'
    exact histogram (``ops.histogram16``) when the caller already has it."""
    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        # int16: the reference's ground() (`array - array.min()`, array_utils.py:102) wraps around in
        # int16 for any frame whose range exceeds 32767, i.e. its own result is an overflow artefact
        raise TypeError("field_centroids_batch needs uint16 frames")
    cnt = x[0].numel()
    hist = ops.histogram16(x) if hist is None else hist
    qs, lo, hi, frac = ops._percentile_plan(cnt, [5, 99.9])

    ranks = np.concatenate([[0, cnt - 1], lo, hi])           # min, max, p-lo ranks, p-hi ranks
    st = ops.order_stats(x, ranks, hist=hist).to(torch.float64)
    vmin, vmax = st[:, 0], st[:, 1]
    gmax = vmax - vmin                                         # max of the grounded frame
    t = torch.as_tensor(frac, dtype=torch.float64, device=x.device)
    a = (st[:, 2:4] - vmin[:, None]) / gmax[:, None]           # normalised lower order statistics
    b = (st[:, 4:6] - vmin[:, None]) / gmax[:, None]
    p = ops.lerp_like_numpy(a, b, t[None, :])                  # [N,2] = (p5, p99.9) of the f64 frame
    thr = (p[:, 1] - p[:, 0]) / 2 + p[:, 0]
    cen = ops.field_cax(x, vmin, gmax, thr)                    # threshold -> fill holes -> centre of mass: row, col, count
    return torch.stack([cen[:, 1], cen[:, 0], cen[:, 2]], dim=1)



def _percentiles_from_hist(x: torch.Tensor, hist: torch.Tensor, q) -> np.ndarray:
    """np.percentile(frame, q) per frame from the exact histogram -> float64 [N, len(q)] on the host"""
    cnt = x[0].numel()
    qs, lo, hi, frac = ops._percentile_plan(cnt, q)
    st = ops.order_stats(x, np.concatenate([lo, hi]), hist=hist).cpu().numpy().astype(np.float64)
    a, b = st[:, : len(qs)], st[:, len(qs):]
    d = b - a
    return np.where((frac >= 0.5)[None, :], b - d * (1 - frac), a + d * frac)


def analyze_batch(frames: torch.Tensor, dpmm: float, bb_diameter_mm: float = 5.0, low_density: bool = False,
                  clean_edges: bool = True):
    """The per-image part of ``WLBaseImage.analyze`` (pylinac/winston_lutz.py:709-725) for a batch of uint16 frames
    resident on the GPU:

        check_inversion_by_histogram((0.01, 50, 99.99))   image.py:899-926
        _clean_edges()                                     winston_lutz.py:1109-1133
        ground(); normalize()                              winston_lutz.py:711-712
        find_field_centroids(is_open_field=False)          winston_lutz.py:764-780
        find_bb_centroids(bb_diameter_mm, low_density)     winston_lutz.py:788-806

    -> dict(record float64 [N, 4] = (field_x, field_y, bb_x, bb_y) in the coordinates of the (possibly edge-cleaned)
    frame, like the reference's points; status int32 [N]: 0 ok, 1 = no BB found (the reference raises ValueError);
    inverted bool [N]; crop int32 [N] = pixels ``_clean_edges`` removed from every side).

    One exact histogram per frame serves every percentile the sequence asks for; the decisions (three comparisons per
    frame) are taken on the host from a [N, 5] table, inversion is applied on the device to the frames that need it.
    A frame whose edges need cleaning changes shape: it is finished on its own (same kernels, batch of one)."""
    from . import decisions, features
    from .roi import rectangle_stats_batch

    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        raise TypeError("analyze_batch needs uint16 frames")
    n, h, w = x.shape
    dev = x.device
    hist = ops.histogram16(x)
    # ---- inversion (|p50 - p0.01| > |p50 - p99.99|): invert -a + max + min, in the frame's dtype
    p = _percentiles_from_hist(x, hist, [0.01, 50, 99.99])
    inverted = np.abs(p[:, 1] - p[:, 0]) > np.abs(p[:, 1] - p[:, 2])
    if inverted.any():
        idx = torch.from_numpy(np.nonzero(inverted)[0]).to(dev)
        x = x.clone()
        xi = x.view(torch.int16)                       # torch has no indexed copies for uint16: same bits as int16
        sub = ops.invert(xi[idx].view(torch.uint16))
        xi[idx] = sub.view(torch.int16)
        hist[idx] = ops.histogram16(sub)
    # ---- edge cleaning decision on the whole batch; frames that need cropping leave the batch
    crop = np.zeros(n, dtype=np.int32)
    record = np.full((n, 4), np.nan, dtype=np.float64)
    status = np.zeros(n, dtype=np.int32)
    keep = np.ones(n, dtype=bool)
    if clean_edges:
        pe = _percentiles_from_hist(x, hist, [5, 99.5])
        ws = 2
        strips = np.array([[0, ws, 0, w], [0, h, 0, ws], [h - ws, h, 0, w], [0, h, w - ws, w]], dtype=np.float64)
        s = rectangle_stats_batch(x, strips)[0].cpu().numpy()                # [N, 4, fields]: 3 = min, 4 = max
        edge_min, edge_max = s[:, :, 3].min(axis=1), s[:, :, 4].max(axis=1)
        rng = pe[:, 1] - pe[:, 0]
        noisy = (edge_min < pe[:, 0] - rng / 10) | (edge_max > pe[:, 1] + rng / 10)
        for i in np.nonzero(noisy)[0]:
            cleaned = decisions.clean_edges(x[i])                              # the reference's loop, one frame
            crop[i] = (h - cleaned.shape[0]) // 2
            one = analyze_batch(cleaned[None], dpmm, bb_diameter_mm, low_density, clean_edges=False)
            record[i], status[i] = one["record"][0], one["status"][0]
            keep[i] = False
    if keep.any():
        sel = x if keep.all() else x.view(torch.int16)[torch.from_numpy(np.nonzero(keep)[0]).to(dev)].view(torch.uint16)
        hsel = hist if keep.all() else hist[torch.from_numpy(np.nonzero(keep)[0]).to(dev)]
        fld = field_centroids_batch(sel, hist=hsel).cpu().numpy()
        bb = features.bb_centroids_batch(sel, dpmm, bb_diameter_mm, low_density=low_density)
        bxy = bb["xy"][:, 0, :].cpu().numpy()
        cnt = bb["count"].cpu().numpy()
        rec = np.concatenate([fld[:, :2], np.where(cnt[:, None] > 0, bxy, np.nan)], axis=1)
        record[keep] = rec
        status[keep] = (cnt == 0).astype(np.int32)
    return dict(record=record, status=status, inverted=inverted, crop=crop)
