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
from . import contrast as _c
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
                raise ValueError("disk ROI is empty" if st == 3 else "disk ROI box exceeds 2**28 pixels")
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


# ------------------------------------------------------------------------------------------- rectangles
def rectangle_vertices(width: float, height: float, center_xy, rotation: float = 0.0) -> np.ndarray:
    """``Rectangle.vertices`` (pylinac/core/geometry.py:692-704) -> float64 [4, 2] (x, y) in tl, tr, br, bl order:
    the half-size square scaled, rotated by ``rotation`` degrees and translated -- the homogeneous matrix product that
    scikit-image's ``EuclideanTransform`` + ``matrix_transform`` evaluate."""
    square = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]])
    scaled = square @ np.diag((width, height)) / 2
    a = np.deg2rad(rotation)
    m = np.array([[np.cos(a), -np.sin(a), center_xy[0]], [np.sin(a), np.cos(a), center_xy[1]], [0, 0, 1]])
    src = np.vstack((scaled[:, 0], scaled[:, 1], np.ones(4)))
    dst = src.T @ m.T
    return dst[:, :2] / dst[:, 2:3]


def _pixels_flat_polygon(vertices_xy: np.ndarray) -> np.ndarray:
    """The (row, col) polygon ``RectangleROI.pixels_flat`` rasterises (roi.py:650-661): bl, br, tr, tl with the
    reference's -1 adjustments."""
    tl, tr, br, bl = vertices_xy
    return np.array([(bl[1] - 1, bl[0]), (br[1] - 1, br[0] - 1), (tr[1], tr[0] - 1), (tl[1], tl[0])], dtype=np.float64)


def polygon_roi_stats_batch(frames: torch.Tensor, polygons_rc) -> tuple[torch.Tensor, torch.Tensor]:
    """Statistics over ``skimage.draw.polygon(r, c, shape=frame.shape)`` pixels -> (float64 [N, K, 6] in STAT_FIELDS
    order, int32 status [N, K]).  ``polygons_rc``: [K, V, 2] (shared) or [N, K, V, 2] (row, col) vertices."""
    f = ops._frames(frames)
    n, h, w = f.shape
    p = polygons_rc if isinstance(polygons_rc, torch.Tensor) else torch.as_tensor(np.asarray(polygons_rc, dtype=np.float64))
    p = p.to(device=f.device, dtype=torch.float64).contiguous()
    if p.ndim == 3:
        k, nv, stride = p.shape[0], p.shape[1], 0
    elif p.ndim == 4 and p.shape[0] == n:
        k, nv, stride = p.shape[1], p.shape[2], p.shape[1] * p.shape[2] * 2
    else:
        raise ValueError("polygons must be [K, V, 2] (shared) or [N, K, V, 2]")
    if p.shape[-1] != 2:
        raise ValueError("each vertex is (row, col)")
    out = torch.empty((n, k, 6), dtype=torch.float64, device=f.device)
    status = torch.empty((n, k), dtype=torch.int32, device=f.device)
    check(_lib.load().pl_polygon_roi_stats(f.data_ptr(), ops._dt(f), n, h, w, p.data_ptr(), nv, k, stride,
                                           out.data_ptr(), status.data_ptr(),
                                           torch.cuda.current_stream(f.device).cuda_stream), "pl_polygon_roi_stats")
    return out, status


def rectangle_roi_stats_batch(frames: torch.Tensor, rects) -> tuple[torch.Tensor, torch.Tensor]:
    """``RectangleROI`` statistics for ``rects`` = [K, 5] rows of (width, height, center x, center y, rotation degrees),
    the same on every frame -> like ``polygon_roi_stats_batch``."""
    r = np.asarray(rects, dtype=np.float64).reshape(-1, 5)
    polys = np.stack([_pixels_flat_polygon(rectangle_vertices(w, h, (cx, cy), rot)) for w, h, cx, cy, rot in r])
    return polygon_roi_stats_batch(frames, polys)


class RectangleROI:
    """pylinac/core/roi.py:481-704 (``Rectangle`` geometry of pylinac/core/geometry.py:634-724 reduced to what the
    statistics need)."""

    def __init__(self, array, width: float, height: float, center, rotation: float = 0.0):
        if width < 2:
            raise ValueError(f"The width must be >= 2. Given {width}")
        if height < 2:
            raise ValueError(f"The height must be >= 2. Given {height}")
        self.width, self.height, self.rotation = width, height, rotation
        self.center = center
        self._xy = (float(center.x), float(center.y)) if hasattr(center, "x") else (float(center[0]), float(center[1]))
        t = array if isinstance(array, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(array))
        if not t.is_cuda:
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        self._array = t
        self._cache = None

    @classmethod
    def from_phantom_center(cls, array, width: float, height: float, angle: float, dist_from_center: float,
                            phantom_center, rotation: float = 0.0):
        """roi.py:484-533"""
        px, py = (phantom_center.x, phantom_center.y) if hasattr(phantom_center, "x") else phantom_center[:2]
        y_shift = np.sin(np.deg2rad(angle)) * dist_from_center
        x_shift = np.cos(np.deg2rad(angle)) * dist_from_center
        return cls(array=array, width=width, height=height, center=(px + x_shift, py + y_shift), rotation=rotation)

    @property
    def area(self) -> float:
        return self.width * self.height

    @property
    def vertices(self) -> np.ndarray:
        """[4, 2] (x, y): tl, tr, br, bl"""
        return rectangle_vertices(self.width, self.height, self._xy, self.rotation)

    tl_corner = property(lambda self: self.vertices[0])
    tr_corner = property(lambda self: self.vertices[1])
    br_corner = property(lambda self: self.vertices[2])
    bl_corner = property(lambda self: self.vertices[3])

    def _s(self) -> np.ndarray:
        if self._cache is None:
            out, status = polygon_roi_stats_batch(self._array[None], _pixels_flat_polygon(self.vertices)[None])
            st = int(status[0, 0])
            if st == 2:
                raise ValueError("rectangle ROI box exceeds 2**28 pixels")
            self._cache = out[0, 0].cpu().numpy()        # status 3 (no pixels): NaNs, like np.mean of an empty array
        return self._cache

    @property
    def pixel_array(self) -> torch.Tensor:
        """roi.py:664-681: the unrotated window ``array[round(tl.y):round(bl.y), round(bl.x):round(br.x)]``."""
        if self.rotation != 0:
            raise ValueError("The pixel array cannot be reshaped into a 2D array when the rotation is not 0.")
        v = self.vertices
        return self._array[int(np.round(v[0][1])):int(np.round(v[3][1])), int(np.round(v[3][0])):int(np.round(v[2][0]))]

    @property
    def pixel_value(self) -> float:
        """The MEAN of the ROI pixels (roi.py:683-686)."""
        return float(self._s()[1])

    mean = pixel_value

    @property
    def std(self) -> float:
        return float(self._s()[2])

    @property
    def min(self) -> float:
        return float(self._s()[3])

    @property
    def max(self) -> float:
        return float(self._s()[4])


# ------------------------------------------------------------------------------------------- contrast ROIs
class LowContrastDiskROI(DiskROI):
    """pylinac/core/roi.py:186-411: a disk ROI with contrast / CNR / visibility against a reference value."""

    def __init__(self, array, radius: float, center, contrast_threshold=None, contrast_reference=None,
                 cnr_threshold=None, contrast_method="Michelson", visibility_threshold: float = 0.1):
        super().__init__(array, radius, center=center)
        self.contrast_threshold = contrast_threshold
        self.cnr_threshold = cnr_threshold
        self.contrast_reference = contrast_reference
        self.contrast_method = contrast_method
        self.visibility_threshold = visibility_threshold

    @classmethod
    def from_phantom_center(cls, array, angle: float, roi_radius: float, dist_from_center: float, phantom_center,
                            contrast_threshold=None, contrast_reference=None, cnr_threshold=None,
                            contrast_method="Michelson", visibility_threshold=0.1):
        base = DiskROI.from_phantom_center(array, angle, roi_radius, dist_from_center, phantom_center)
        return cls(base._array, roi_radius, base.center, contrast_threshold, contrast_reference, cnr_threshold,
                   contrast_method, visibility_threshold)

    @property
    def diameter(self) -> float:
        return self.radius * 2

    @property
    def _contrast_array(self) -> np.ndarray:
        return np.array((self.pixel_value, self.contrast_reference))

    @property
    def signal_to_noise(self) -> float:
        return float(np.array(self.pixel_value) / self.std)

    @property
    def contrast(self) -> float:
        return _c.contrast(self._contrast_array, self.contrast_method)

    @property
    def contrast_to_noise(self) -> float:
        return float(np.array(self.contrast) / self.std)

    @property
    def michelson(self) -> float:
        return _c.michelson(self._contrast_array)

    @property
    def weber(self) -> float:
        return _c.weber(feature=self.pixel_value, background=self.contrast_reference)

    @property
    def rms(self) -> float:
        return _c.rms(self._contrast_array)

    @property
    def visibility(self) -> float:
        return _c.visibility(array=self._contrast_array, radius=self.radius, std=self.std, algorithm=self.contrast_method)

    @property
    def cnr_constant(self) -> float:
        return self.contrast_to_noise * self.diameter

    @property
    def contrast_constant(self) -> float:
        return self.contrast * self.diameter

    @property
    def passed(self) -> bool:
        return self.contrast > self.contrast_threshold

    @property
    def passed_visibility(self) -> bool:
        return self.visibility > self.visibility_threshold

    @property
    def passed_contrast_constant(self) -> bool:
        return self.contrast_constant > self.contrast_threshold

    @property
    def passed_cnr_constant(self) -> bool:
        return self.cnr_constant > self.cnr_threshold


class HighContrastDiskROI(DiskROI):
    """pylinac/core/roi.py:414-478: a disk ROI read through its ``max`` / ``min``."""

    def __init__(self, array, radius: float, center, contrast_threshold: float):
        super().__init__(array=array, radius=radius, center=center)
        self.contrast_threshold = contrast_threshold

    @classmethod
    def from_phantom_center(cls, array, angle: float, roi_radius: float, dist_from_center: float, phantom_center,
                            contrast_threshold: float):
        base = DiskROI.from_phantom_center(array, angle, roi_radius, dist_from_center, phantom_center)
        return cls(base._array, roi_radius, base.center, contrast_threshold)


class ThicknessROI(RectangleROI):
    """pylinac/ct.py:300-313: the slice-thickness wire ramp of the CatPhan HU module -- Gaussian(1) of the unrotated
    rectangle window, maximum along its short axis, FWHM of the resulting profile."""

    @property
    def long_profile(self):
        from .profile import FWXMProfile

        win = self.pixel_array.contiguous()
        smooth = ops.gaussian_filter(win[None], 1)                         # image.load(pixel_array).filter(1, "gaussian")
        axis = int(np.argmin(win.shape))
        prof = ops.reduce_axis(smooth, axis, "max")[0].cpu().numpy().astype(_NP_OF[win.dtype])
        return FWXMProfile(values=prof)

    @property
    def wire_fwhm(self) -> float:
        return self.long_profile.field_width_px


_NP_OF = {torch.uint8: np.uint8, torch.uint16: np.uint16, torch.int16: np.int16, torch.int32: np.int32,
          torch.int64: np.int64, torch.float32: np.float32, torch.float64: np.float64}
