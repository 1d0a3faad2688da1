"""Mirror of the peak / FWXM part of ``pylinac.core.profile``:

  find_peaks        pylinac/core/profile.py:2545-2623   (+ _parse_peak_args :2626-2649)
  MultiProfile      pylinac/core/profile.py:2002-2176
  FWXMProfile       pylinac/core/profile.py:578-611 on ProfileBase :195-344

Same names, arguments, return formats and error behaviour; the peak search itself runs in the
batched HIP kernel (csrc/peaks.hip).  ``find_peaks_batch``/``fwxm_batch`` are the device-resident
batch forms used by the pipelines.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass

import numpy as np
import torch

from . import array_utils as au
from . import ops

LEFT = "left"
RIGHT = "right"


class Normalization(enum.Enum):
    """pylinac/core/profile.py:170-176."""

    NONE = None
    GEOMETRIC_CENTER = "Geometric center"
    BEAM_CENTER = "Beam center"
    MAX = "Max"


@dataclass
class Point:
    """The two fields of ``pylinac.core.geometry.Point`` that the profile classes use."""

    idx: int | None = None
    value: float | None = None
    x: float = 0
    y: float = 0
    z: float = 0


def _to_device_profile(values) -> torch.Tensor:
    if isinstance(values, torch.Tensor):
        t = values
        if not t.is_cuda:
            t = t.to(au._device())
    else:
        a = np.asarray(values)
        au.array_not_empty(a)
        if a.ndim != 1:
            raise ValueError(f"Array was multidimensional. Must pass 1D array; found {a.ndim}")
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(au._device())
    return t.to(torch.float64)


def find_peaks(
    values,
    threshold=-np.inf,
    peak_separation=0,
    max_number=None,
    fwxm_height: float = 0.5,
    min_width: int = 0,
    search_region=(0.0, 1.0),
    peak_sort: str = "prominences",
    required_prominence=None,
):
    """Mirror of ``pylinac.core.profile.find_peaks`` -> ``(peak_idxs, peak_props)``.

    Known, documented divergence: where several peaks tie EXACTLY on the sort key (or on height
    inside the ``distance`` filter) the reference's order comes from ``np.argsort``'s default
    introsort and is implementation-defined; this backend uses the stable order."""
    x = _to_device_profile(values)
    res = ops.find_peaks_batch(
        x, threshold=threshold, peak_separation=peak_separation, max_number=max_number,
        fwxm_height=fwxm_height, min_width=min_width, search_region=search_region,
        peak_sort=peak_sort, required_prominence=required_prominence,
    )
    return res.to_host(0)


class MultiProfile:
    """pylinac/core/profile.py:2002-2176."""

    def __init__(self, values):
        self.values = values
        self.peaks = []
        self.valleys = []

    def __len__(self):
        return len(self.values)

    def __getitem__(self, items):
        return self.values[items]

    def normalize(self, norm_val=None) -> None:
        self.values = au.normalize(np.asarray(self.values), value=None if norm_val == "max" else norm_val)

    def ground(self) -> float:
        v = np.asarray(self.values)
        mn = v.min()
        self.values = au.ground(v)
        return mn

    def filter(self, size=0.05, kind: str = "median") -> None:
        self.values = au.filter(np.asarray(self.values), size=size, kind=kind)

    def find_peaks(self, threshold=0.3, min_distance=0.05, max_number=None,
                   search_region=(0.0, 1.0), peak_sort="prominences"):
        """profile.py:2050-2103 -> (indices, values)."""
        peak_idxs, peak_props = find_peaks(
            self.values, threshold=threshold, peak_separation=min_distance, max_number=max_number,
            search_region=search_region, peak_sort=peak_sort,
        )
        self.peaks = [Point(value=v, idx=i) for i, v in zip(peak_idxs, peak_props["peak_heights"])]
        return peak_idxs, peak_props["peak_heights"]

    def find_valleys(self, threshold=0.3, min_distance=0.05, max_number=None,
                     search_region=(0.0, 1.0)):
        """profile.py:2105-2133: peaks of ``-values``."""
        values = np.asarray(self.values)
        valley_idxs, _ = find_peaks(
            -values, threshold=threshold, peak_separation=min_distance, max_number=max_number,
            search_region=search_region,
        )
        self.valleys = [Point(value=values[i], idx=i) for i in valley_idxs]
        return valley_idxs, values[valley_idxs]

    def find_fwxm_peaks(self, threshold=0.3, min_distance=0.05, max_number=None,
                        search_region=(0.0, 1.0), peak_sort="prominences", required_prominence=None):
        """profile.py:2135-2176: FWXM centre ``int(round(lt + (rt - lt) / 2))`` (banker's rounding)."""
        values = np.asarray(self.values)
        _, peak_props = find_peaks(
            values, threshold=threshold, peak_separation=min_distance, max_number=max_number,
            search_region=search_region, peak_sort=peak_sort, required_prominence=required_prominence,
        )
        idxs = [int(round(lt + (rt - lt) / 2))
                for lt, rt in zip(peak_props["left_ips"], peak_props["right_ips"])]
        vals = [values[i] for i in idxs]
        self.peaks = [Point(value=v, idx=i) for i, v in zip(idxs, vals)]
        return np.array(idxs), np.array(vals)


def _linear_at(xp: np.ndarray, fp: np.ndarray, x):
    """k=1, s=0 ``UnivariateSpline`` == piecewise-linear interpolation with linear extrapolation
    (profile.py:249-262)."""
    x = np.asarray(x, dtype=float)
    i = np.clip(np.searchsorted(xp, x, side="right") - 1, 0, len(xp) - 2)
    t = (x - xp[i]) / (xp[i + 1] - xp[i])
    return fp[i] + t * (fp[i + 1] - fp[i])


class FWXMProfile:
    """pylinac/core/profile.py:578-611 on ProfileBase (:195-344): a profile with one large signal
    whose edges are the FWXM intersections of its most prominent peak."""

    def __init__(self, values, x_values=None, ground: bool = False,
                 normalization=Normalization.NONE, fwxm_height: float = 50):
        values = np.asarray(values)
        if values.ndim > 1:
            raise ValueError(f"Array was multidimensional. Must pass 1D array; found {values.ndim}")
        self.fwxm_height = fwxm_height
        if x_values is None:
            x_values = np.arange(len(values))
        x_values = np.asarray(x_values)
        x_diff = np.diff(x_values)
        if x_diff.max() > 0 > x_diff.min():
            raise ValueError("X values must be monotonically increasing or decreasing")
        sort_idxs = np.argsort(x_values)
        self.x_values = x_values[sort_idxs]
        self.values = values[sort_idxs]
        self._cache = {}
        if ground:
            self.values = au.ground(self.values)
        if isinstance(normalization, str):
            normalization = Normalization(normalization)
        if normalization == Normalization.MAX:
            self.normalize()
        elif normalization == Normalization.GEOMETRIC_CENTER:
            self.normalize(au.geometric_center_value(self.values))
        elif normalization == Normalization.BEAM_CENTER:
            self.normalize(self.y_at_x(self.center_idx))

    def __len__(self):
        return len(self.values)

    def __getitem__(self, items):
        return self.values[items]

    def normalize(self, norm_val=None) -> None:
        self.values = au.normalize(self.values, value=norm_val)
        self._cache.clear()

    def x_at_x_idx(self, x):
        r = _linear_at(np.arange(len(self.x_values), dtype=float), self.x_values.astype(float), x)
        return float(r) if r.size == 1 else r

    def y_at_x(self, x):
        r = _linear_at(self.x_values.astype(float), np.asarray(self.values, dtype=float), x)
        return float(r) if r.size == 1 else r

    def _edges(self):
        if "edges" not in self._cache:
            _, props = find_peaks(self.values, fwxm_height=self.fwxm_height / 100, max_number=1)
            # IndexError when no peak exists -- same as the reference (profile.py:608)
            self._cache["edges"] = (props["left_ips"][0], props["right_ips"][0])
        return self._cache["edges"]

    def field_edge_idx(self, side: str) -> float:
        left, right = self._edges()
        return self.x_at_x_idx(left if side == LEFT else right)

    @property
    def center_idx(self) -> float:
        left = self.field_edge_idx(LEFT)
        right = self.field_edge_idx(RIGHT)
        return abs(right - left) / 2 + left

    @property
    def field_width_px(self) -> float:
        left = self.field_edge_idx(LEFT)
        right = self.field_edge_idx(RIGHT)
        return max(right, left) - min(right, left)


class CircleProfile(MultiProfile):
    """pylinac/core/profile.py:2179-2402: a profile sampled along a circle
    (``ndimage.map_coordinates(image, [y, x], order=0)``)."""

    def __init__(self, center, radius: float, image_array, start_angle=0, ccw: bool = True,
                 sampling_ratio: float = 1.0):
        self.center = center if hasattr(center, "x") else Point(x=center[0], y=center[1])
        self.radius = radius
        image_array = np.asarray(image_array)
        self._ensure_array_size(image_array, self.radius + self.center.x, self.radius + self.center.y)
        self.image_array = image_array
        self.start_angle = start_angle
        self.ccw = ccw
        self.sampling_ratio = sampling_ratio
        super().__init__(self._profile)

    @staticmethod
    def _ensure_array_size(array, min_width: float, min_height: float) -> None:
        """profile.py:2394-2402 (only the +x / +y extents are checked, like the reference)."""
        if array.shape[1] < min_width or array.shape[0] < min_height:
            raise ValueError("Array size not large enough to compute profile")

    @property
    def _radii(self):
        return np.array([self.radius], dtype=float)

    @property
    def _divisor(self) -> float:
        return 1.0

    @property
    def size(self) -> float:
        return np.pi * max(self._radii) * 2 * self.sampling_ratio

    @property
    def _radians(self):
        return ops.circle_radians(self.size, self.start_angle, self.ccw)

    @property
    def x_locations(self):
        return np.cos(self._radians) * self.radius + self.center.x

    @property
    def y_locations(self):
        return np.sin(self._radians) * self.radius + self.center.y

    @property
    def _profile(self):
        s = au._Staged(self.image_array)
        out = ops.circle_profile(s.t, self.center.x, self.center.y, self._radii, self.size,
                                 self.start_angle, self.ccw, self._divisor)[0].cpu().numpy()
        if self._divisor == 1.0:  # map_coordinates returns the image dtype
            return out.astype(self.image_array.dtype)
        return out


class CollapsedCircleProfile(CircleProfile):
    """pylinac/core/profile.py:2405-2483: mean of ``num_profiles`` concentric circle profiles."""

    def __init__(self, center, radius: float, image_array, start_angle=0, ccw: bool = True,
                 sampling_ratio: float = 1.0, width_ratio: float = 0.1, num_profiles: int = 20):
        if not 0 <= width_ratio <= 1:
            raise ValueError("width_ratio must be within (0, 1)")
        self.width_ratio = width_ratio
        self.num_profiles = num_profiles
        super().__init__(center, radius, image_array, start_angle, ccw, sampling_ratio)

    @property
    def _radii(self):
        return np.linspace(start=self.radius * (1 - self.width_ratio),
                           stop=self.radius * (1 + self.width_ratio), num=self.num_profiles)

    @property
    def _divisor(self) -> float:
        return float(self.num_profiles)


# ------------------------------------------------------------------------ device-resident batch
FWXM_FIELDS = ("n_peaks", "peak_idx", "peak_height", "prominence", "left_edge", "right_edge",
               "center", "width")


def fwxm_batch(profiles: torch.Tensor, fwxm_height: float = 50) -> torch.Tensor:
    """FWXMProfile edges/centre/width for every row of ``profiles`` [N,L] -> float64 [N, 8]
    (FWXM_FIELDS; NaN where a profile has no peak).  Stays on the device."""
    res = ops.find_peaks_batch(profiles, cap=1, fwxm_height=fwxm_height / 100, max_number=1)
    return ops.fwxm_record(res)
