"""ROI statistics (SURVEY.md section 8 "next" row f3): device mirror of ``pylinac.core.roi.DiskROI``.

``DiskROI`` keeps the reference's constructor, ``from_phantom_center`` and the ``pixel_value`` (median) / ``mean`` /
``std`` / ``min`` / ``max`` properties (pylinac/core/roi.py:38-140); ``disk_roi_stats_batch`` /
``rectangle_stats_batch`` are the batch forms used after phantom localisation (pylinac/ct.py:554-586: HU,
uniformity and low-contrast ROIs of every slice).  Pixel membership is ``skimage.draw.disk``'s.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

STAT_FIELDS = ("count", "mean", "std", "min", "max", "median")


def _stats(frames: torch.Tensor, rois: torch.Tensor, kind: int) -> tuple[torch.Tensor, torch.Tensor]:
    f = ops._frames(frames)
    n, h, w = f.shape
    r = rois.to(device=f.device, dtype=torch.float64).contiguous()
    if r.ndim == 2:
        k, stride = r.shape[0], 0                       # the same ROIs on every frame
    elif r.ndim == 3 and r.shape[0] == n:
        k, stride = r.shape[1], r.shape[1] * 4
    else:
        raise ValueError("rois must be [K, 4] (shared) or [N, K, 4]")
    if r.shape[-1] != 4:
        raise ValueError("each ROI is four numbers")
    out = torch.empty((n, k, 6), dtype=torch.float64, device=f.device)
    status = torch.empty((n, k), dtype=torch.int32, device=f.device)
    check(_lib.load().pl_roi_stats(f.data_ptr(), ops._dt(f), n, h, w, r.data_ptr(), k, stride, kind, out.data_ptr(),
                                   status.data_ptr(), torch.cuda.current_stream(f.device).cuda_stream), "pl_roi_stats")
    return out, status


def disk_roi_stats_batch(frames: torch.Tensor, centers_xy, radius) -> tuple[torch.Tensor, torch.Tensor]:
    """Statistics of disk ROIs -> (float64 [N, K, 6] in STAT_FIELDS order, int32 status [N, K]).
    ``centers_xy``: [K, 2] (shared by all frames) or [N, K, 2] (x, y) pixel coordinates; ``radius``: scalar or [K]."""
    c = torch.as_tensor(np.asarray(centers_xy, dtype=np.float64)) if not isinstance(centers_xy, torch.Tensor) else centers_xy
    c = c.to(torch.float64)
    rad = torch.as_tensor(np.broadcast_to(np.asarray(radius, dtype=np.float64), c.shape[:-1]).copy()).to(c.device)
    rois = torch.cat([c, rad[..., None], torch.zeros_like(rad)[..., None]], dim=-1)
    return _stats(frames, rois, 0)


def rectangle_stats_batch(frames: torch.Tensor, boxes_r0r1c0c1) -> tuple[torch.Tensor, torch.Tensor]:
    """Statistics of ``frame[r0:r1, c0:c1]`` windows -> same layout as ``disk_roi_stats_batch``."""
    b = boxes_r0r1c0c1 if isinstance(boxes_r0r1c0c1, torch.Tensor) else torch.as_tensor(np.asarray(boxes_r0r1c0c1, dtype=np.float64))
    return _stats(frames, b, 1)


class DiskROI:
    """pylinac/core/roi.py:38-140 (``Circle`` geometry reduced to ``center`` / ``radius``)."""

    def __init__(self, array, radius: float, center):
        self.radius = radius
        self.center = center
        self._xy = (float(center.x), float(center.y)) if hasattr(center, "x") else (float(center[0]), float(center[1]))
        t = array if isinstance(array, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(array))
        if not t.is_cuda:
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        self._array = t
        self._cache = None

    @classmethod
    def from_phantom_center(cls, array, angle: float, roi_radius: float, dist_from_center: float, phantom_center):
        """roi.py:41-68, 91-101."""
        px, py = (phantom_center.x, phantom_center.y) if hasattr(phantom_center, "x") else phantom_center[:2]
        y_shift = np.sin(np.deg2rad(angle)) * dist_from_center
        x_shift = np.cos(np.deg2rad(angle)) * dist_from_center
        return cls(array=array, center=(px + x_shift, py + y_shift), radius=roi_radius)

    def _s(self) -> np.ndarray:
        if self._cache is None:
            out, status = disk_roi_stats_batch(self._array[None], [self._xy], self.radius)
            st = int(status[0, 0])
            if st == 1:   # the reference indexes without a shape: negative indices wrap, large ones raise
                raise IndexError("disk ROI leaves the image")
            if st:
                raise ValueError("disk ROI is empty or larger than 16384 pixels")
            self._cache = out[0, 0].cpu().numpy()
        return self._cache

    @property
    def pixel_value(self) -> float:
        """The median pixel value of the ROI."""
        return float(self._s()[5])

    @property
    def mean(self) -> float:
        return float(self._s()[1])

    @property
    def std(self) -> float:
        return float(self._s()[2])

    @property
    def min(self) -> float:
        return float(self._s()[3])

    @property
    def max(self) -> float:
        return float(self._s()[4])
