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
``((a - min) / max') >= t`` in float64 (``pl_scaled_binary``).  Image inversion and edge cleaning
(``check_inversion_by_histogram``, ``_clean_edges``; winston_lutz.py:709-710) change the frame
shape per image and stay with the caller in this round.
"""
from __future__ import annotations

import torch

from . import ops


def field_centroids_batch(frames: torch.Tensor) -> torch.Tensor:
    """-> float64 [N, 3] = (x, y, filled_pixel_count) of the field centroid of every frame."""
    x = ops._frames(frames)
    if x.dtype != torch.uint16:
        # int16: the reference's ground() (`array - array.min()`, array_utils.py:102) wraps around in
        # int16 for any frame whose range exceeds 32767, i.e. its own result is an overflow artefact
        raise TypeError("field_centroids_batch needs uint16 frames")
    cnt = x[0].numel()
    hist = ops.histogram16(x)
    qs, lo, hi, frac = ops._percentile_plan(cnt, [5, 99.9])
    import numpy as np

    ranks = np.concatenate([[0, cnt - 1], lo, hi])           # min, max, p-lo ranks, p-hi ranks
    st = ops.order_stats(x, ranks, hist=hist).to(torch.float64)
    vmin, vmax = st[:, 0], st[:, 1]
    gmax = vmax - vmin                                         # max of the grounded frame
    t = torch.as_tensor(frac, dtype=torch.float64, device=x.device)
    a = (st[:, 2:4] - vmin[:, None]) / gmax[:, None]           # normalised lower order statistics
    b = (st[:, 4:6] - vmin[:, None]) / gmax[:, None]
    p = ops.lerp_like_numpy(a, b, t[None, :])                  # [N,2] = (p5, p99.9) of the f64 frame
    thr = (p[:, 1] - p[:, 0]) / 2 + p[:, 0]
    binary = ops.scaled_binary(x, vmin, gmax, thr)
    filled = ops.fill_holes(binary, connectivity_bg=4)
    cen = ops.binary_centroid(filled)                          # row, col, count
    return torch.stack([cen[:, 1], cen[:, 0], cen[:, 2]], dim=1)
