"""Percentile-driven image decisions: the small yes/no questions the analyzers ask of a
frame before the main analysis, batched over [N, H, W] 16-bit frames.  The order statistics, extrema and axis sums come
from the device (exact histogram selection, ``pl_minmax``, ``pl_reduce_axis``, ``pl_roi_stats``); what is left per
frame is a handful of scalar comparisons, written exactly as the reference writes them.

  has_noise          PFDicomImage._has_noise            pylinac/picketfence.py:229-238
  pf_orientation     PicketFence.orientation            pylinac/picketfence.py:1501-1526
  corners_inverted   BaseImage.check_inversion          pylinac/core/image.py:868-897
  clean_edges        WLBaseImage._clean_edges           pylinac/winston_lutz.py:1109-1133
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .roi import rectangle_stats_batch


def has_noise(frames: torch.Tensor) -> np.ndarray:
    """-> bool [N]: max > 1.25 * p99.5, or min < 0.75 * p0.5 and |min - p0.5| > 0.1 * (p99.5 - p0.5)."""
    x = ops._frames(frames)
    mn, mx = (t.cpu().numpy() for t in ops.minmax(x))
    p = ops.percentile(x, [0.5, 99.5]).numpy()
    near_min, near_max = p[:, 0], p[:, 1]
    max_is_extreme = mx > near_max * 1.25
    min_is_extreme = (mn < near_min * 0.75) & (np.abs(mn - near_min) > 0.1 * (near_max - near_min))
    return max_is_extreme | min_is_extreme


def pf_orientation(frames: torch.Tensor) -> list[str]:
    """-> "Left-Right" / "Up-Down" per frame.  Pixels below the median are raised to it (the float median is cast
    into the integer frame, so for integer frames this is ``max(a, trunc(median))``), then the spread between the
    85th and 99th percentile of the column sums and of the row sums decides."""
    x = ops._frames(frames)
    n, h, w = x.shape
    cnt = h * w
    st = ops.order_stats(x, [(cnt - 1) // 2, cnt // 2]).cpu().numpy().astype(np.float64)
    median = (st[:, 0] + st[:, 1]) / 2                              # np.median: mean of the two middle values
    info = np.iinfo(np.uint16 if x.dtype == torch.uint16 else np.int16)
    out = []
    for i in range(n):                                             # the clip level differs per frame
        temp = ops.clip(x[i:i + 1], float(np.trunc(median[i])), float(info.max))
        row_sum = ops.reduce_axis(temp, 0, "sum")[0].cpu().numpy()
        col_sum = ops.reduce_axis(temp, 1, "sum")[0].cpu().numpy()
        row80, row90 = np.percentile(row_sum, [85, 99])
        col80, col90 = np.percentile(col_sum, [85, 99])
        out.append("Left-Right" if (row90 - row80) < (col90 - col80) else "Up-Down")
    return out


def corners_inverted(frames: torch.Tensor, box_size: int = 20, position=(0.0, 0.0)) -> np.ndarray:
    """-> bool [N]: mean of the four corner boxes above the frame mean (the reference then inverts the frame)."""
    x = ops._frames(frames)
    n, h, w = x.shape
    row_pos = max(int(position[0] * h), 1)
    col_pos = max(int(position[1] * w), 1)
    boxes = np.array([[row_pos, row_pos + box_size, col_pos, col_pos + box_size],
                      [h - row_pos - box_size, h - row_pos, col_pos, col_pos + box_size],
                      [row_pos, row_pos + box_size, w - col_pos - box_size, w - col_pos],
                      [h - row_pos - box_size, h - row_pos, w - col_pos - box_size, w - col_pos]], dtype=np.float64)
    stats, status = rectangle_stats_batch(x, boxes)
    if int(status.abs().sum()):
        raise ValueError("corner boxes do not fit inside the frame")
    s = stats.cpu().numpy()
    avg = (s[:, :, 1] * s[:, :, 0]).sum(axis=1) / s[:, :, 0].sum(axis=1)        # mean over the four equal boxes
    total = ops.reduce_axis(x, 1, "sum").cpu().numpy().sum(axis=1) / (h * w)
    return avg > total


def clean_edges(frame: torch.Tensor, window_size: int = 2) -> torch.Tensor:
    """One frame [H, W] -> the cropped frame: while an edge strip of ``window_size`` pixels holds a value more than
    10 % of the (p5 .. p99.5) range outside it, crop ``window_size`` pixels from every side (at most min(shape)/10
    times)."""
    x = frame
    safety_stop = min(x.shape) / 10
    while safety_stop > 0:
        xc = x.contiguous()[None]
        p = ops.percentile(xc, [5, 99.5]).numpy()[0]
        near_min, near_max = p[0], p[1]
        img_range = near_max - near_min
        h, w = x.shape
        ws = window_size
        strips = np.array([[0, ws, 0, w], [0, h, 0, ws], [h - ws, h, 0, w], [0, h, w - ws, w]], dtype=np.float64)
        s = rectangle_stats_batch(xc, strips)[0].cpu().numpy()[0]
        edge_min, edge_max = s[:, 3].min(), s[:, 4].max()
        if not (edge_min < (near_min - img_range / 10) or edge_max > (near_max + img_range / 10)):
            break
        x = x[ws:-ws, ws:-ws]                                         # BaseImage.crop(window_size)
        safety_stop -= 1
    return x.contiguous()
