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
import math
import warnings
from dataclasses import dataclass
from functools import cached_property

import numpy as np
import torch

from . import array_utils as au
from . import ops
from .geometry import Circle, Point  # noqa: F401  (Point is re-exported: analyzers import it from here)

LEFT = "left"
RIGHT = "right"


class Normalization(enum.Enum):
    """pylinac/core/profile.py:170-176."""

    NONE = None
    GEOMETRIC_CENTER = "Geometric center"
    BEAM_CENTER = "Beam center"
    MAX = "Max"


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
    if max_number is not None and max_number < 0:
        # the reference slices `[::-1][:max_number]`: a negative count would DROP that many of the smallest peaks
        raise ValueError("max_number must be None or >= 0")
    x = _to_device_profile(values)
    if max_number == 0:                                  # `[::-1][:0]`: no peaks (profile.py:2616-2623)
        res = ops.find_peaks_batch(x, threshold=threshold, peak_separation=peak_separation, max_number=None,
                                   fwxm_height=fwxm_height, min_width=min_width, search_region=search_region,
                                   peak_sort=peak_sort, required_prominence=required_prominence)
        idx, props = res.to_host(0)
        return idx[:0], {k: v[:0] for k, v in props.items()}
    res = ops.find_peaks_batch(
        x, threshold=threshold, peak_separation=peak_separation, max_number=max_number,
        fwxm_height=fwxm_height, min_width=min_width, search_region=search_region,
        peak_sort=peak_sort, required_prominence=required_prominence,
    )
    return res.to_host(0)


def stretch(array, min: int = 0, max: int = 1, fill_dtype=None):
    """pylinac/core/profile.py:40-83, the deprecated profile-module ``stretch``: (array - min) / range scaled to ``max`` (or to
    ``fill_dtype``'s maximum, then cast); the new minimum is NOT applied (the reference's line is commented out)."""
    warnings.warn("Using stretch from the profile module is deprecated. Use 'stretch' from the pylinac.core.array_utils module",
                  DeprecationWarning)
    new_max = max
    if fill_dtype is not None:
        new_max = (np.iinfo(fill_dtype) if np.issubdtype(fill_dtype, np.integer) else np.finfo(fill_dtype)).max
    out = au.stretch(np.asarray(array), min=0, max=1) * new_max          # ground / range on the device, then the scale
    return out.astype(fill_dtype) if fill_dtype else out


class ProfileMixin:
    """pylinac/core/profile.py:86-153: in-place manipulations of a profile's ``values`` through ``array_utils`` (device
    kernels); every 1-D profile class carries them."""

    def invert(self) -> None:
        self.values = au.invert(np.asarray(self.values))

    def bit_invert(self) -> None:
        self.values = au.bit_invert(np.asarray(self.values))

    def normalize(self, norm_val=None) -> None:
        self.values = au.normalize(np.asarray(self.values), value=None if isinstance(norm_val, str) and norm_val == "max" else norm_val)

    def stretch(self, min: float = 0, max: float = 1) -> None:
        self.values = au.stretch(np.asarray(self.values), min=min, max=max)

    def convert_to_dtype(self, dtype) -> None:
        self.values = au.convert_to_dtype(np.asarray(self.values), dtype=dtype)

    def ground(self) -> float:
        v = np.asarray(self.values)
        mn = v.min()
        self.values = au.ground(v)
        return mn

    def filter(self, size: float = 0.05, kind: str = "median") -> None:
        self.values = au.filter(np.asarray(self.values), size=size, kind=kind)

    def __len__(self):
        return len(self.values)

    def __getitem__(self, items):
        return self.values[items]


class MultiProfile(ProfileMixin):
    """pylinac/core/profile.py:2002-2176."""

    def __init__(self, values):
        self.values = values
        self.peaks = []
        self.valleys = []

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


class ProfileBase(ProfileMixin):
    """pylinac/core/profile.py:195-576 (``ProfileBase``): sorting by x, grounding / normalisation, index <-> position
    look-ups (a k=1, s=0 ``UnivariateSpline`` is piecewise-linear interpolation), centre / width from the subclass's
    ``field_edge_idx``, the metric plug-in protocol (``compute``)."""

    def __init__(self, values, x_values=None, ground: bool = False, normalization=Normalization.NONE,
                 interpolation_order: int = 1):
        values = np.asarray(values)
        if values.ndim > 1:
            raise ValueError(f"Array was multidimensional. Must pass 1D array; found {values.ndim}")
        if interpolation_order != 1:
            raise NotImplementedError("index <-> position look-ups are built for interpolation_order=1 (the reference's default; "
                                      "no analyzer passes another)")
        self.metrics = []
        self.metric_values = {}
        self._interp_order = interpolation_order
        if x_values is None:
            x_values = np.arange(len(values))
        x_values = np.asarray(x_values)
        x_diff = np.diff(x_values)
        if x_diff.max() > 0 > x_diff.min():
            raise ValueError("X values must be monotonically increasing or decreasing")
        sort_idxs = np.argsort(x_values)
        self._cache = {}
        self.x_values = x_values[sort_idxs]
        self.values = values[sort_idxs]
        if ground:
            self.values = au.ground(self.values)
        normalization = _enum(normalization, Normalization)
        if normalization == Normalization.MAX:
            self.normalize()
        elif normalization == Normalization.GEOMETRIC_CENTER:
            self.normalize(au.geometric_center_value(self.values))
        elif normalization == Normalization.BEAM_CENTER:
            self.normalize(self.y_at_x(self.center_idx))

    # ``values`` is a plain attribute in the reference and every mutator rebinds it; here the rebinding also drops what was
    # derived from the old samples (the edge search, the smoothed derivative): ``field_edge_idx`` searches the current
    # values on every call like the reference's, while ``center_idx`` & co. stay what they were at first use
    # (``cached_property`` there and here)
    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, v) -> None:
        self._values = v
        self._cache.clear()

    # the same for ``x_values``: ``PhysicalProfileMixin.gamma`` shifts the x-values of deep copies (profile.py:861-866), and what
    # the edge search derived from the old coordinates (the cubic through the smoothed derivative) must not answer for the new.
    # An IN-PLACE edit of either array is invisible to this: rebind the attribute, as every mutator of the reference does.
    @property
    def x_values(self):
        return self._x_values

    @x_values.setter
    def x_values(self, v) -> None:
        self._x_values = v
        self._cache.clear()

    def x_at_x(self, x):
        """profile.py:242-247: deprecated alias"""
        warnings.warn("x_at_x is deprecated. Use x_at_x_idx instead", DeprecationWarning)
        return self.x_at_x_idx(x)

    def compute(self, metrics):
        """profile.py:531-575: the profile-metric plug-in protocol.  ``metrics``: one object or an iterable of objects with
        ``inject_profile(profile)``, ``calculate()`` and ``full_name`` -- the reference's own ``ProfileMetric`` subclasses
        (pylinac/metrics/profile.py) run unchanged on these profiles.  One metric -> its value, several -> a dict."""
        from .image import _uniquify

        if hasattr(metrics, "calculate"):
            metrics = [metrics]
        values = {}
        for metric in metrics:
            metric.inject_profile(self)
            self.metrics.append(metric)
            key = _uniquify(list(values.keys()) + list(self.metric_values.keys()), metric.full_name)
            values[key] = metric.calculate()
        self.metric_values |= values
        return values[key] if len(values) == 1 else values

    def x_at_x_idx(self, x):
        r = _linear_at(np.arange(len(self.x_values), dtype=float), self.x_values.astype(float), x)
        return float(r) if r.size == 1 else r

    def x_idx_at_x(self, x: float) -> int:
        """profile.py:269-271: index of the x-value closest to ``x``"""
        return int(np.argmin(np.abs(self.x_values - x)))

    def y_at_x(self, x):
        r = _linear_at(self.x_values.astype(float), np.asarray(self.values, dtype=float), x)
        return float(r) if r.size == 1 else r

    def x_at_y(self, y, side: str):
        """profile.py:279-292: the x-value where one flank of the profile reaches ``y`` (linear ``interp1d`` over the
        samples left / right of the centre, sorted by value like scipy does for unsorted abscissae)."""
        s_idx = self.x_idx_at_x(self.center_idx)
        vals = np.asarray(self.values, dtype=float)
        xs = np.asarray(self.x_values, dtype=float)
        v, x = (vals[:s_idx], xs[:s_idx]) if side == LEFT else (vals[s_idx:], xs[s_idx:])
        order = np.argsort(v, kind="mergesort")
        r = np.asarray(_Linear1d(v[order], x[order], extrapolate=False)(y))
        return float(r) if r.size == 1 else r

    def field_edge_idx(self, side: str) -> float:
        raise NotImplementedError

    def field_x_values(self, in_field_ratio: float) -> np.ndarray:
        """profile.py:308-321: the x-values inside the central ``in_field_ratio`` of the field, inclusive of the edges
        (floor / ceil of the two bounds)."""
        margin = (1 - in_field_ratio) / 2 * self.field_width_px               # what each edge gives up
        ends = (self.field_edge_idx(side=LEFT) + margin, self.field_edge_idx(side=RIGHT) - margin)
        keep = (self.x_values >= math.floor(min(ends))) & (self.x_values <= math.ceil(max(ends)))
        return self.x_values[np.nonzero(keep)[0]]

    def field_values(self, in_field_ratio: float = 0.8) -> np.ndarray:
        """profile.py:345-352"""
        return np.atleast_1d(self.y_at_x(self.field_x_values(in_field_ratio)))

    def field_indices(self, in_field_ratio: float) -> tuple:
        """profile.py:299-306 -> (left, right, width) in x-value units"""
        xs = self.field_x_values(in_field_ratio)
        left, right = xs[0], xs[-1]
        return left, right, max(right, left) - min(right, left)

    def _resample_kwargs(self) -> dict:
        """constructor arguments a resampled copy keeps (the subclasses' own parameters)"""
        return {}

    def as_resampled(self, interpolation_factor: float = 10, order: int = 3):
        """profile.py:353-390: ``scipy.ndimage.zoom(values, interpolation_factor, order=3, mode="nearest",
        grid_mode=False)`` on the device, x-values re-spaced over the same range, a new profile of this class."""
        import warnings

        if order != 3:
            raise NotImplementedError("as_resampled is built for the cubic spline (order=3) the reference defaults to")
        values = np.asarray(self.values)
        arr_range = values.max() - values.min()
        if values.dtype != float and arr_range < 100:
            warnings.warn(f"Array range is small ({arr_range}) and is not a float. Interpolation may look step-like. "
                          "Consider converting the array to a float before passing it to this method.", UserWarning)
        new_y = ops.zoom1d_cubic(_to_device_profile(values.astype(np.float64)), interpolation_factor).cpu().numpy()
        if values.dtype.kind in "iu":     # scipy writes into an array of the input dtype: round half away, clamp
            info = np.iinfo(values.dtype)
            new_y = np.clip(np.where(new_y > 0, new_y + 0.5, new_y - 0.5), info.min, info.max).astype(values.dtype)
        elif values.dtype == np.float32:
            new_y = new_y.astype(np.float32)
        new_x = np.linspace(self.x_values.min(), self.x_values.max(), len(new_y))
        return type(self)(values=new_y, x_values=new_x, ground=False, normalization=Normalization.NONE,
                          **self._resample_kwargs())

    def resample_to(self, target_profile: "ProfileBase"):
        """profile.py:392-431: this profile's values linearly interpolated at the target's x-values (no
        extrapolation), as a new profile of this class."""
        target_x = np.asarray(target_profile.x_values, dtype=float)
        self_x = np.asarray(self.x_values, dtype=float)
        if target_x.min() < self_x.min() or target_x.max() > self_x.max():
            raise ValueError(
                "The target profile x-values are outside this profiles range. Extrapolation is not allowed. "
                f"self x-values: {self_x.min()} to {self_x.max()}. target x-values: {target_x.min()} to {target_x.max()}. ")
        xs = torch.from_numpy(np.ascontiguousarray(self_x))
        dev = _to_device_profile(np.asarray(self.values, dtype=float))
        target_y = ops.interp1d(xs.to(dev.device), dev, torch.from_numpy(np.ascontiguousarray(target_x)).to(dev.device),
                                kind="linear").reshape(-1).cpu().numpy()
        return type(self)(values=target_y, x_values=target_x)

    @cached_property
    def center_idx(self) -> float:
        """profile.py:322-327 (a ``cached_property`` there too: fixed at first use)"""
        left = self.field_edge_idx(LEFT)
        right = self.field_edge_idx(RIGHT)
        return abs(right - left) / 2 + left

    @cached_property
    def geometric_center_idx(self) -> float:
        """profile.py:329-332"""
        return self.x_at_x_idx(au.geometric_center_idx(self.values))

    @cached_property
    def cax_index(self) -> float:
        """profile.py:334-337"""
        return self.x_at_x_idx((len(self.x_values) - 1) / 2)

    @cached_property
    def field_width_px(self) -> float:
        """profile.py:339-344"""
        left = self.field_edge_idx(LEFT)
        right = self.field_edge_idx(RIGHT)
        return max(right, left) - min(right, left)


_ProfileBase = ProfileBase          # (the name earlier rounds used)


class FWXMProfile(ProfileBase):
    """pylinac/core/profile.py:578-611: a profile with one large signal whose edges are the FWXM intersections of its
    most prominent peak."""

    def __init__(self, values, x_values=None, ground: bool = False,
                 normalization=Normalization.NONE, fwxm_height: float = 50):
        self.fwxm_height = fwxm_height
        super().__init__(values, x_values=x_values, ground=ground, normalization=normalization)

    def _resample_kwargs(self) -> dict:
        return dict(fwxm_height=self.fwxm_height)

    def _edges(self):
        if "edges" not in self._cache:
            _, props = find_peaks(self.values, fwxm_height=self.fwxm_height / 100, max_number=1)
            # IndexError when no peak exists -- same as the reference (profile.py:608)
            self._cache["edges"] = (props["left_ips"][0], props["right_ips"][0])
        return self._cache["edges"]

    def field_edge_idx(self, side: str) -> float:
        left, right = self._edges()
        return self.x_at_x_idx(left if side == LEFT else right)


class _CubicOnHost:
    """The not-a-knot cubic spline ``interp1d(x, y, kind="cubic")`` interpolates with, evaluated point by point on the
    host from second derivatives the device solved for (``ops.cubic_spline_moments``).  ``bounds_error=True`` like
    scipy's default: a query outside the knots raises ValueError."""

    def __init__(self, x: np.ndarray, y: np.ndarray, moments: np.ndarray):
        self.x, self.y, self.m = np.asarray(x, float), np.asarray(y, float), np.asarray(moments, float)

    def __call__(self, xq):
        v = np.atleast_1d(np.asarray(xq, dtype=float))
        if (v < self.x[0]).any():
            raise ValueError("A value in x_new is below the interpolation range.")
        if (v > self.x[-1]).any():
            raise ValueError("A value in x_new is above the interpolation range.")
        hi = np.clip(np.searchsorted(self.x, v, side="left"), 1, len(self.x) - 1)
        lo = hi - 1
        h = self.x[hi] - self.x[lo]
        a, b = self.x[hi] - v, v - self.x[lo]
        out = ((self.m[lo] * a ** 3 + self.m[hi] * b ** 3) / (6.0 * h) + (self.y[lo] / h - self.m[lo] * h / 6.0) * a
               + (self.y[hi] / h - self.m[hi] * h / 6.0) * b)
        return out if np.ndim(xq) else out.reshape(())


class InflectionDerivativeProfile(ProfileBase):
    """pylinac/core/profile.py:612-680: field edges = the extrema of the derivative of the Gaussian-smoothed profile,
    refined on its cubic interpolant.  Smoothing, gradient and the spline solve run on the device; the two
    one-dimensional BFGS refinements are the reference's own ``scipy.optimize.minimize`` calls on the host (SURVEY.md
    section 8 row f4)."""

    def __init__(self, values, x_values=None, ground: bool = False, normalization=Normalization.NONE,
                 edge_smoothing_ratio: float = 0.003):
        self.edge_smoothing_ratio = edge_smoothing_ratio
        super().__init__(values, x_values=x_values, ground=ground, normalization=normalization)

    def _resample_kwargs(self) -> dict:
        return dict(edge_smoothing_ratio=self.edge_smoothing_ratio)

    def _derivative(self):
        if "diff" not in self._cache:
            v = _to_device_profile(np.asarray(self.values, dtype=float))
            sm = ops.gaussian_filter1d(v[None], self.edge_smoothing_ratio * len(self.values))[0]
            d1 = ops.gradient1d(sm)
            xs = torch.from_numpy(np.ascontiguousarray(self.x_values, dtype=np.float64)).to(d1.device)
            m = ops.cubic_spline_moments(xs, d1)[0]
            diff = d1.cpu().numpy()
            self._cache["diff"] = (diff, _CubicOnHost(self.x_values, diff, m.cpu().numpy()))
        return self._cache["diff"]

    def _inflection_edge(self, side: str) -> float:
        from scipy.optimize import minimize

        key = "infl_" + side
        if key not in self._cache:
            diff, f_diff = self._derivative()
            if side == LEFT:
                initial_guess = self.x_at_x_idx(np.argmax(diff))
                self._cache[key] = minimize(lambda x: -f_diff(x), x0=initial_guess).x[0]
            else:
                initial_guess = self.x_at_x_idx(np.argmin(diff))
                self._cache[key] = minimize(f_diff, x0=initial_guess).x[0]
        return self._cache[key]

    def field_edge_idx(self, side: str) -> float:
        """profile.py:656-670"""
        return self._inflection_edge(side)


class HillProfile(InflectionDerivativeProfile):
    """pylinac/core/profile.py:682-740: a Hill function fitted to a window about each inflection edge."""

    def __init__(self, values, x_values=None, ground: bool = False, normalization=Normalization.NONE,
                 edge_smoothing_ratio: float = 0.003, hill_window_ratio: float = 0.1):
        self.hill_window_ratio = hill_window_ratio
        super().__init__(values, x_values=x_values, ground=ground, normalization=normalization,
                         edge_smoothing_ratio=edge_smoothing_ratio)

    def _resample_kwargs(self) -> dict:
        return dict(edge_smoothing_ratio=self.edge_smoothing_ratio, hill_window_ratio=self.hill_window_ratio)

    def field_edge_idx(self, side: str) -> float:
        """profile.py:708-728"""
        left_infl, right_infl = self._inflection_edge(LEFT), self._inflection_edge(RIGHT)
        window_size = (right_infl - left_infl) * self.hill_window_ratio
        mid = left_infl if side == LEFT else right_infl
        left_idx, right_idx = self.x_idx_at_x(mid - window_size), self.x_idx_at_x(mid + window_size)
        fit = Hill.fit(x_data=self.x_values[left_idx:right_idx + 1], y_data=self.values[left_idx:right_idx + 1])
        return fit.inflection_idx()["index (exact)"]


class PhysicalProfileMixin:
    """pylinac/core/profile.py:742-775: profiles whose x-values are pixels of known size (``dpmm``) or already physical
    positions (``dpmm=None``: the mean spacing is the implicit dots-per-mm)."""

    def _init_physical(self, dpmm: float | None) -> None:
        self.dpmm = dpmm
        self.implicit_dpmm = np.mean(np.diff(self.x_values)) if dpmm is None else dpmm

    @property
    def physical_x_values(self) -> np.ndarray:
        if self.dpmm is None:
            return self.x_values
        return self.x_values / self.dpmm + 0.5 / self.dpmm          # half-pixel offset

    @property
    def field_width_mm(self) -> float:
        return self.field_width_px / self.implicit_dpmm

    def gamma(self, evaluation_profile, dose_to_agreement: float = 3, distance_to_agreement: float = 3,
              gamma_cap_value: float = 2, dose_threshold: float = 5, fill_value: float = np.nan, return_profiles: bool = False):
        """profile.py:822-874: geometric gamma of this (reference) profile against ``evaluation_profile``, both shifted so that
        their geometric centres sit at 0, on their physical x-values (``gamma.gamma_geometric`` -> ``pl_gamma_geometric``).
        ``return_profiles`` -> (gamma, the shifted copy of self, the shifted copy of the evaluation profile)."""
        import copy

        from .gamma import gamma_geometric

        if not isinstance(evaluation_profile, PhysicalProfileMixin):
            raise ValueError("The evaluation profile must also be a physical profile.")
        reference, evaluation = copy.deepcopy(self), copy.deepcopy(evaluation_profile)
        reference.x_values = reference.x_values - reference.geometric_center_idx
        evaluation.x_values = evaluation.x_values - evaluation.geometric_center_idx
        g = gamma_geometric(reference=np.asarray(reference.values), reference_coordinates=reference.physical_x_values,
                            evaluation=np.asarray(evaluation.values), evaluation_coordinates=evaluation.physical_x_values,
                            dose_to_agreement=dose_to_agreement, distance_to_agreement=distance_to_agreement,
                            gamma_cap_value=gamma_cap_value, dose_threshold=dose_threshold, fill_value=fill_value)
        return (g, reference, evaluation) if return_profiles else g

    def as_simple_profile(self):
        """profile.py:936-948: the non-physical class over the physical x-values"""
        return type(self).__bases__[-1](values=self.values, x_values=self.physical_x_values)

    def as_resampled(self, interpolation_resolution_mm: float = 0.1, order: int = 3, grid: bool = True):
        """profile.py:950-1011: zoom by ``1 / (dpmm * resolution)`` (``grid_mode=grid``: every sample is a cell of
        physical size), x-values widened by the half-cell offset, a new physical profile at ``1 / resolution`` dpmm."""
        import warnings

        if order != 3:
            raise NotImplementedError("as_resampled is built for the cubic spline (order=3) the reference defaults to")
        values = np.asarray(self.values)
        arr_range = values.max() - values.min()
        if values.dtype != float and arr_range < 100:
            warnings.warn(f"Array range is small ({arr_range}) and is not a float. Interpolation may look step-like. "
                          "Consider converting the array to a float before passing it to this method.", UserWarning)
        factor = 1 / (self.dpmm * interpolation_resolution_mm)
        new_y = ops.zoom1d_cubic(_to_device_profile(values.astype(np.float64)), factor, grid_mode=grid).cpu().numpy()
        if values.dtype.kind in "iu":
            info = np.iinfo(values.dtype)
            new_y = np.clip(np.where(new_y > 0, new_y + 0.5, new_y - 0.5), info.min, info.max).astype(values.dtype)
        elif values.dtype == np.float32:
            new_y = new_y.astype(np.float32)
        if grid:
            offset = 0.5 - 1 / (2 * factor)
            new_x = np.linspace(self.x_values.min() - offset, self.x_values.max() + offset, len(new_y))
        else:
            new_x = np.linspace(self.x_values.min(), self.x_values.max(), len(new_y))
        return type(self)(values=new_y, x_values=new_x, dpmm=factor * self.dpmm, ground=False,
                          normalization=Normalization.NONE, **self._resample_kwargs())


class FWXMProfilePhysical(PhysicalProfileMixin, FWXMProfile):
    """profile.py:1014-1043"""

    def __init__(self, values, dpmm: float | None = None, x_values=None, ground: bool = False,
                 normalization=Normalization.NONE, fwxm_height: float = 50):
        FWXMProfile.__init__(self, values=values, x_values=x_values, ground=ground, normalization=normalization,
                             fwxm_height=fwxm_height)
        self._init_physical(dpmm)


class InflectionDerivativeProfilePhysical(PhysicalProfileMixin, InflectionDerivativeProfile):
    """profile.py:1046-1081"""

    def __init__(self, values, dpmm: float | None = None, x_values=None, ground: bool = False,
                 normalization=Normalization.NONE, edge_smoothing_ratio: float = 0.003):
        InflectionDerivativeProfile.__init__(self, values=values, x_values=x_values, ground=ground,
                                             normalization=normalization, edge_smoothing_ratio=edge_smoothing_ratio)
        self._init_physical(dpmm)


class HillProfilePhysical(PhysicalProfileMixin, HillProfile):
    """profile.py:1084-1115"""

    def __init__(self, values, dpmm: float | None = None, x_values=None, ground: bool = False,
                 normalization=Normalization.NONE, edge_smoothing_ratio: float = 0.003, hill_window_ratio: float = 0.1):
        HillProfile.__init__(self, values=values, x_values=x_values, ground=ground, normalization=normalization,
                             edge_smoothing_ratio=edge_smoothing_ratio, hill_window_ratio=hill_window_ratio)
        self._init_physical(dpmm)


class CircleProfile(MultiProfile, Circle):
    """pylinac/core/profile.py:2179-2402: a profile sampled along a circle
    (``ndimage.map_coordinates(image, [y, x], order=0)``); a ``Circle`` too (``center``, ``radius``, ``diameter``, ``area``)."""

    def __init__(self, center, radius: float, image_array, start_angle=0, ccw: bool = True,
                 sampling_ratio: float = 1.0):
        Circle.__init__(self, center, radius)
        image_array = np.asarray(image_array)
        self._ensure_array_size(image_array, self.radius + self.center.x, self.radius + self.center.y)
        self.image_array = image_array
        self.start_angle = start_angle
        self.ccw = ccw
        self.sampling_ratio = sampling_ratio
        self._x_locations = self._y_locations = None
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

    # x_locations / y_locations are cached properties in the reference, and ``roll`` replaces them
    @property
    def x_locations(self):
        if self._x_locations is None:
            self._x_locations = np.cos(self._radians) * self.radius + self.center.x
        return self._x_locations

    @x_locations.setter
    def x_locations(self, v):
        self._x_locations = v

    @property
    def y_locations(self):
        if self._y_locations is None:
            self._y_locations = np.sin(self._radians) * self.radius + self.center.y
        return self._y_locations

    @y_locations.setter
    def y_locations(self, v):
        self._y_locations = v

    def find_peaks(self, threshold=0.3, min_distance=0.05, max_number=None, search_region=(0.0, 1.0)):
        """profile.py:2284-2296: MultiProfile.find_peaks + the peaks' image coordinates."""
        out = super().find_peaks(threshold, min_distance, max_number, search_region)
        self._map_peaks()
        return out

    def find_valleys(self, threshold=0.3, min_distance=0.05, max_number=None, search_region=(0.0, 1.0)):
        """profile.py:2298-2310 (like the reference, it is ``self.peaks`` that gets mapped)."""
        out = super().find_valleys(threshold, min_distance, max_number, search_region)
        self._map_peaks()
        return out

    def find_fwxm_peaks(self, threshold=0.3, min_distance=0.05, max_number=None, search_region=(0.0, 1.0)):
        """profile.py:2312-2324"""
        out = super().find_fwxm_peaks(threshold, min_distance, max_number, search_region=search_region)
        self._map_peaks()
        return out

    def _map_peaks(self) -> None:
        """profile.py:2326-2329"""
        for peak in self.peaks:
            peak.x = self.x_locations[int(peak.idx)]
            peak.y = self.y_locations[int(peak.idx)]

    def roll(self, amount: int) -> None:
        """profile.py:2331-2340: roll the profile and its x / y coordinates to the left."""
        self.values = np.roll(self.values, -amount)
        self.x_locations = np.roll(self.x_locations, -amount)
        self.y_locations = np.roll(self.y_locations, -amount)

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


# ------------------------------------------------------------------------------ SingleProfile (a11)
class Interpolation(enum.Enum):
    """pylinac/core/profile.py:162-167."""

    NONE = None
    LINEAR = "Linear"
    SPLINE = "Spline"


class Edge(enum.Enum):
    """pylinac/core/profile.py:179-184."""

    FWHM = "FWHM"
    INFLECTION_DERIVATIVE = "Inflection Derivative"
    INFLECTION_HILL = "Inflection Hill"


class Centering(enum.Enum):
    """pylinac/core/profile.py:187-192."""

    MANUAL = "Manual"
    BEAM_CENTER = "Beam center"
    GEOMETRIC_CENTER = "Geometric center"


def _enum(value, cls):
    """pylinac.core.utilities.convert_to_enum; a member of the HOST package's enum of the same name (``pylinac.Edge.FWHM`` handed
    to these classes by the reference's analyzers) converts through its value."""
    if isinstance(value, cls):
        return value
    return cls(value.value if isinstance(value, enum.Enum) else value)


class _Linear1d:
    """``scipy.interpolate.interp1d(x, y)`` (linear) for the scalar look-ups SingleProfile makes on the host:
    ``np.interp`` when not extrapolating (what scipy itself dispatches to), otherwise scipy's slope form ``(y_hi - y_lo) / (x_hi - x_lo) * (x_new - x_lo) + y_lo`` with
    ``hi = clip(searchsorted(x, x_new), 1, len - 1)``; out of range -> ValueError unless ``extrapolate``."""

    def __init__(self, x, y, extrapolate: bool):
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        self.extrapolate = extrapolate

    def __call__(self, x_new):
        xn = np.asarray(x_new, dtype=np.float64)
        if not self.extrapolate:
            if (xn < self.x[0]).any():
                raise ValueError(f"A value ({xn.min()}) in x_new is below the interpolation range's minimum value ({self.x[0]}).")
            if (xn > self.x[-1]).any():
                raise ValueError(f"A value ({xn.max()}) in x_new is above the interpolation range's maximum value ({self.x[-1]}).")
            # scipy hands a non-extrapolating linear interp1d over float64 / int data to np.interp (interp1d.__init__:
            # ``_call_linear_np``), whose result on a grid point is the sample itself
            return np.interp(xn, self.x, self.y)
        hi = np.clip(np.searchsorted(self.x, xn), 1, len(self.x) - 1).astype(int)
        lo = hi - 1
        slope = (self.y[hi] - self.y[lo]) / (self.x[hi] - self.x[lo])
        return slope * (xn - self.x[lo]) + self.y[lo]


class Hill:
    """pylinac/core/hill.py:11-65: the four-parameter sigmoid ``a + (b - a) / (1 + (c / x) ** d)`` fitted to a
    penumbra window.  ``fit`` calls ``scipy.optimize.curve_fit`` with the reference's start values."""

    params: np.ndarray

    @staticmethod
    def _func(x, a, b, c, d):
        return a + (b - a) / (1.0 + (c / x) ** d)

    @classmethod
    def fit(cls, x_data: np.ndarray, y_data: np.ndarray) -> "Hill":
        from scipy.optimize import curve_fit

        params, _ = curve_fit(cls._func, x_data, y_data, p0=(min(y_data), max(y_data), np.median(x_data), 0))
        return cls.from_params(params)

    @classmethod
    def from_params(cls, params) -> "Hill":
        inst = cls()
        inst.params = params
        return inst

    def inflection_idx(self) -> dict:
        idx = self.params[2] * math.pow((self.params[3] - 1) / (self.params[3] + 1), 1 / self.params[3])
        return {"index (exact)": idx, "index (rounded)": int(round(idx))}

    def gradient_at(self, x: float) -> float:
        cxd = math.pow(self.params[2] / x, self.params[3])
        return (self.params[1] - self.params[0]) * self.params[3] * cxd / (math.pow(cxd + 1, 2) * x)

    def x(self, y: float) -> float:
        return self.params[2] * math.pow((y - self.params[0]) / (self.params[1] - y), 1 / self.params[3])

    def y(self, x: float) -> float:
        return self.params[0] + (self.params[1] - self.params[0]) / (1 + (self.params[2] / x) ** self.params[3])


class SingleProfile(ProfileMixin):
    """pylinac/core/profile.py:1118-1633: a profile with one large signal (a beam profile).

    Same constructor arguments, dictionary keys and error behaviour as the reference for the FWHM,
    INFLECTION_DERIVATIVE and INFLECTION_HILL edge methods.  Resampling (``pl_interp1d``), grounding / normalisation (elementwise kernels) and
    the FWXM search (``pl_find_peaks``) run on the GPU; ``values`` is the host copy the reference's users
    read, ``values_device`` the resident tensor.  The handful of scalar look-ups and three-parameter fits of
    ``field_data`` are host numpy, like the reference's.
    """

    def __init__(self, values, dpmm: float | None = None, interpolation=Interpolation.LINEAR, ground: bool = True,
                 interpolation_resolution_mm: float = 0.1, interpolation_factor: float = 10,
                 normalization_method=Normalization.BEAM_CENTER, edge_detection_method=Edge.FWHM,
                 edge_smoothing_ratio: float = 0.003, hill_window_ratio: float = 0.1, x_values=None,
                 centering=Centering.BEAM_CENTER):
        self._interp_method = _enum(interpolation, Interpolation)
        self._interpolation_res = interpolation_resolution_mm
        self._interpolation_factor = interpolation_factor
        self._norm_method = _enum(normalization_method, Normalization)
        self._edge_method = _enum(edge_detection_method, Edge)
        self._edge_smoothing_ratio = edge_smoothing_ratio
        self._hill_window_ratio = hill_window_ratio
        self._centering = _enum(centering, Centering)
        self.dpmm = dpmm
        dev_values = _to_device_profile(values)
        fitted, new_dpmm, x_indices = self._interpolate(dev_values, x_values, dpmm, interpolation_resolution_mm,
                                                        interpolation_factor, self._interp_method)
        self.x_indices = x_indices
        self._x_interp1d = _Linear1d(np.arange(len(x_indices)), x_indices, extrapolate=False)
        self._ground = ground
        if ground:
            fitted = ops.ground(fitted[None, None])[0, 0]           # fitted_values -= fitted_values.min()
        self._set_values(fitted)
        norm = self._normalize(fitted, self._norm_method)
        self._set_values(norm)

    def _set_values(self, t: torch.Tensor) -> None:
        self.values_device = t
        self._values = t.cpu().numpy()
        self._y_interp1d = _Linear1d(self.x_indices, self._values, extrapolate=True)

    # ``values`` is what the reference's users read and what the ProfileMixin methods rebind (invert, stretch, filter ...);
    # a rebinding refreshes the resident copy the searches run on.  Like the reference, it does NOT rebuild ``_y_interp1d``
    # (the look-up function of the constructor's values, profile.py:1213-1215).
    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, v) -> None:
        self._values = np.asarray(v)
        self.values_device = _to_device_profile(self._values)

    def resample(self, interpolation_factor: int = 10, interpolation_resolution_mm: float = 0.1) -> "SingleProfile":
        """profile.py:1283-1304: a new profile of the current values at another resolution"""
        return SingleProfile(values=self.values, x_values=self.x_indices, dpmm=(1 / self._interpolation_res) if self.dpmm else None,
                             interpolation=self._interp_method, ground=self._ground,
                             interpolation_resolution_mm=interpolation_resolution_mm, interpolation_factor=interpolation_factor,
                             normalization_method=self._norm_method, edge_detection_method=self._edge_method,
                             edge_smoothing_ratio=self._edge_smoothing_ratio, hill_window_ratio=self._hill_window_ratio)

    def gamma(self, evaluation_profile: "SingleProfile", distance_to_agreement: int = 1, dose_to_agreement: float = 1,
              gamma_cap_value: float = 2, dose_threshold: float = 5, global_dose: bool = True, fill_value: float = np.nan):
        """profile.py:1939-1993: 1-D gamma of this (reference) profile against ``evaluation_profile`` on their physical
        x-indices (``pl_gamma1d``)."""
        from .gamma import gamma_1d

        if not self.dpmm or not evaluation_profile.dpmm:
            raise ValueError("At least one profile does not have the dpmm attribute. Physical spacing cannot be determined. "
                             "Set it before performing gamma analysis.")
        return gamma_1d(reference=self.values, evaluation=evaluation_profile.values, reference_coordinates=self.x_indices,
                        evaluation_coordinates=evaluation_profile.x_indices, dose_to_agreement=dose_to_agreement,
                        distance_to_agreement=distance_to_agreement, gamma_cap_value=gamma_cap_value, global_dose=global_dose,
                        dose_threshold=dose_threshold, fill_value=fill_value)[0]

    # --- interpolation (profile.py:1306-1360)
    @staticmethod
    def _interpolate(values: torch.Tensor, x_values, dpmm, interpolation_resolution, interpolation_factor,
                     interp_method: Interpolation):
        n = values.numel()
        if x_values is None:
            x_values = np.array(range(n))
        x_values = np.asarray(x_values)
        if np.diff(x_values).min() < 0:
            raise ValueError("Profile values must be monotonically increasing")
        if interp_method == Interpolation.NONE:
            return values, dpmm, x_values
        if dpmm is not None:
            samples = int(round(len(x_values) / (dpmm * interpolation_resolution)))
            new_dpmm = 1 / interpolation_resolution
        else:
            samples = int(round(len(x_values) * interpolation_factor))
            new_dpmm = None
        resampling_factor = samples / n
        offset = 0.5 - 1 / (2 * resampling_factor)
        new_x = np.linspace(x_values[0] - offset, x_values[-1] + offset, num=samples)
        dev = values.device
        xs = torch.from_numpy(np.ascontiguousarray(x_values, dtype=np.float64)).to(dev)
        xq = torch.from_numpy(new_x).to(dev)
        kind = "linear" if interp_method == Interpolation.LINEAR else "cubic"
        return ops.interp1d(xs, values, xq, kind=kind), new_dpmm, new_x

    def _normalize(self, values: torch.Tensor, method: Normalization) -> torch.Tensor:
        """profile.py:1362-1371."""
        if method == Normalization.NONE:
            return values
        if method == Normalization.MAX:
            return ops.normalize(values[None, None])[0, 0]
        if method == Normalization.GEOMETRIC_CENTER:
            norm = self._geometric_center(self.values)["value (exact)"]
        else:
            norm = self.beam_center()["value (@rounded)"]
        return ops.normalize(values[None, None], float(norm))[0, 0]

    # --- scalar look-ups (profile.py:1217-1235)
    def _x_interp_to_original(self, location):
        x = self._x_interp1d(location)
        if isinstance(location, (float, int)) or np.size(location) == 1:
            return float(np.asarray(x).reshape(-1)[0])
        return x

    def _y_original_to_interp(self, location):
        y = self._y_interp1d(location)
        if isinstance(location, (float, int)) or np.size(location) == 1:
            return float(np.asarray(y).reshape(-1)[0])
        return y

    def _geometric_center(self, values) -> dict:
        """profile.py:1373-1385."""
        return {"index (exact)": self._x_interp_to_original(au.geometric_center_idx(values)),
                "value (exact)": au.geometric_center_value(values)}

    def geometric_center(self) -> dict:
        return self._geometric_center(self.values)

    def beam_center(self) -> dict:
        """profile.py:1390-1409."""
        if self._edge_method == Edge.FWHM:
            data = self.fwxm_data(x=50)
            return {"index (rounded)": data["center index (rounded)"], "index (exact)": data["center index (exact)"],
                    "value (@rounded)": data["center value (@rounded)"]}
        infl = self.inflection_data()
        mid_point = infl["left index (exact)"] + (infl["right index (exact)"] - infl["left index (exact)"]) / 2
        return {"index (rounded)": int(round(mid_point)), "index (exact)": mid_point,
                "value (@rounded)": self._y_original_to_interp(int(round(mid_point)))}

    def inflection_data(self) -> dict:
        """profile.py:1635-1670 (INFLECTION_DERIVATIVE): edges = outermost extrema of the gradient of the
        Gaussian-smoothed profile.  Smoothing, gradient and both peak searches run on the device."""
        if self._edge_method == Edge.FWHM:
            raise ValueError("FWHM edge method does not have inflection points. Use a different edge detection method")
        sm = ops.gaussian_filter1d(self.values_device[None], self._edge_smoothing_ratio * len(self.values))[0]
        d1 = ops.gradient1d(sm)
        peak_idxs, _ = find_peaks(d1, threshold=0.8, peak_separation=0.05)        # MultiProfile(d1).find_peaks
        valley_idxs, _ = find_peaks(-d1, threshold=0.8, peak_separation=0.05)     # MultiProfile(d1).find_valleys
        left_idx = self._x_interp_to_original(peak_idxs[0])
        right_idx = self._x_interp_to_original(valley_idxs[-1])
        if self._edge_method == Edge.INFLECTION_HILL:
            # profile.py:1675-1721: a Hill function fitted to a window about each derivative extremum.  The window is
            # a few dozen samples and the fit four parameters: the reference's own per-profile optimiser
            # (scipy.optimize.curve_fit = MINPACK lmdif) runs on the host, like the "top" fit of field_data
            # (a device Levenberg-Marquardt only pays off for IC-Profiler-scale batches).
            half = int(round(self._hill_window_ratio * abs(right_idx - left_idx) / 2))
            x_left = np.array([x for x in np.arange(left_idx - half, left_idx + half) if x >= 0])
            left_hill = Hill.fit(x_left, self._y_original_to_interp(x_left))
            x_right = np.array([x for x in np.arange(right_idx - half, right_idx + half) if x < d1.numel()])
            right_hill = Hill.fit(x_right, self._y_original_to_interp(x_right))
            left_infl, right_infl = left_hill.inflection_idx(), right_hill.inflection_idx()
            return {
                "left index (rounded)": left_infl["index (rounded)"],
                "left index (exact)": left_infl["index (exact)"],
                "right index (rounded)": right_infl["index (rounded)"],
                "right index (exact)": right_infl["index (exact)"],
                "left value (@exact)": left_hill.y(left_infl["index (exact)"]),
                "right value (@exact)": right_hill.y(right_infl["index (exact)"]),
                "left Hill params": left_hill.params,
                "right Hill params": right_hill.params,
            }
        return {
            "left index (rounded)": int(round(left_idx)),
            "left index (exact)": left_idx,
            "right index (rounded)": int(round(right_idx)),
            "right index (exact)": right_idx,
            "left value (@rounded)": self._y_original_to_interp(int(round(left_idx))),
            "left value (@exact)": self._y_original_to_interp(left_idx),
            "right value (@rounded)": self._y_original_to_interp(int(round(right_idx))),
            "right value (@exact)": self._y_original_to_interp(right_idx),
        }

    def penumbra(self, lower: int = 20, upper: int = 80) -> dict:
        """profile.py:1723-1908: penumbra positions / widths (and, for the Hill method, the edge gradients), with the
        reference's keys for each edge method."""
        if lower > upper:
            raise ValueError("Upper penumbra value must be larger than the lower penumbra value")
        if self._edge_method == Edge.FWHM:
            upper_data, lower_data = self.fwxm_data(x=upper), self.fwxm_data(x=lower)
            data = {
                f"left {lower}% index (exact)": lower_data["left index (exact)"],
                f"left {lower}% value (@rounded)": lower_data["left value (@rounded)"],
                f"left {upper}% index (exact)": upper_data["left index (exact)"],
                f"left {upper}% value (@rounded)": upper_data["left value (@rounded)"],
                f"right {lower}% index (exact)": lower_data["right index (exact)"],
                f"right {lower}% value (@rounded)": lower_data["right value (@rounded)"],
                f"right {upper}% index (exact)": upper_data["right index (exact)"],
                f"right {upper}% value (@rounded)": upper_data["right value (@rounded)"],
                "left values": self.values[lower_data["left index (rounded)"]:upper_data["left index (rounded)"]],
                "right values": self.values[upper_data["right index (rounded)"]:lower_data["right index (rounded)"]],
                "left penumbra width (exact)": abs(upper_data["left index (exact)"] - lower_data["left index (exact)"]),
                "right penumbra width (exact)": abs(upper_data["right index (exact)"] - lower_data["right index (exact)"]),
            }
        elif self._edge_method == Edge.INFLECTION_DERIVATIVE:
            infl = self.inflection_data()
            vmax = self.values.max()
            lower_left = self.fwxm_data(x=max(infl["left value (@exact)"] / vmax * lower / 50 * 100, 1))
            upper_left = self.fwxm_data(x=min(infl["left value (@exact)"] / vmax * upper / 50 * 100, 99))
            lower_right = self.fwxm_data(x=max(infl["right value (@exact)"] / vmax * lower / 50 * 100, 1))
            upper_right = self.fwxm_data(x=min(infl["right value (@exact)"] / vmax * upper / 50 * 100, 99))
            data = {
                f"left {lower}% index (exact)": lower_left["left index (exact)"],
                f"left {upper}% index (exact)": upper_left["left index (exact)"],
                f"right {lower}% index (exact)": lower_right["right index (exact)"],
                f"right {upper}% index (exact)": upper_right["right index (exact)"],
                "left values": self._y_original_to_interp(
                    np.arange(lower_left["left index (rounded)"], upper_left["left index (rounded)"])),
                "right values": self._y_original_to_interp(
                    np.arange(upper_right["right index (rounded)"], lower_right["right index (rounded)"])),
                "left penumbra width (exact)": abs(upper_left["left index (exact)"] - lower_left["left index (exact)"]),
                "right penumbra width (exact)": abs(upper_right["right index (exact)"] - lower_right["right index (exact)"]),
            }
        else:
            infl = self.inflection_data()
            left_hill = Hill.from_params(infl["left Hill params"])
            right_hill = Hill.from_params(infl["right Hill params"])
            lower_left_value = infl["left value (@exact)"] * lower / 50
            upper_left_value = infl["left value (@exact)"] * upper / 50
            lower_right_value = infl["right value (@exact)"] * lower / 50
            upper_right_value = infl["right value (@exact)"] * upper / 50
            lower_left_index, upper_left_index = left_hill.x(lower_left_value), left_hill.x(upper_left_value)
            lower_right_index, upper_right_index = right_hill.x(lower_right_value), right_hill.x(upper_right_value)
            data = {
                f"left {lower}% index (exact)": lower_left_index,
                f"left {lower}% value (exact)": lower_left_value,
                f"left {upper}% index (exact)": upper_left_index,
                f"left {upper}% value (exact)": upper_left_value,
                f"right {lower}% index (exact)": lower_right_index,
                f"right {lower}% value (exact)": lower_right_value,
                f"right {upper}% index (exact)": upper_right_index,
                f"right {upper}% value (exact)": upper_right_value,
                "left values": self.values[int(round(lower_left_index)):int(round(upper_left_index))],
                "right values": self.values[int(round(upper_right_index)):int(round(lower_right_index))],
                "left penumbra width (exact)": abs(upper_left_index - lower_left_index),
                "right penumbra width (exact)": abs(upper_right_index - lower_right_index),
                "left gradient (exact)": left_hill.gradient_at(infl["left index (exact)"]),
                "right gradient (exact)": right_hill.gradient_at(infl["right index (exact)"]),
            }
            if self.dpmm:
                data["left gradient (exact) %/mm"] = data["left gradient (exact)"] * self.dpmm * 100
                data["right gradient (exact) %/mm"] = data["right gradient (exact)"] * self.dpmm * 100
        if self.dpmm:
            data["left penumbra width (exact) mm"] = data["left penumbra width (exact)"] / self.dpmm
            data["right penumbra width (exact) mm"] = data["right penumbra width (exact)"] / self.dpmm
        return data

    def fwxm_data(self, x: int = 50) -> dict:
        """profile.py:1411-1461.  The slice of ``x_indices`` by ROUNDED PHYSICAL positions for
        "field values" is the reference's own (negative positions slice from the end)."""
        if not 0 <= x <= 100:
            raise ValueError("x must be within (0, 100)")
        _, peak_props = find_peaks(self.values_device, fwxm_height=x / 100, max_number=1)
        left_idx = float(self._x_interp_to_original(peak_props["left_ips"][0]))
        right_idx = float(self._x_interp_to_original(peak_props["right_ips"][0]))
        width = right_idx - left_idx
        fwxm_center_idx = (right_idx - left_idx) / 2 + left_idx
        data = {
            "width (exact)": width,
            "width (rounded)": int(round(width)),
            "center index (rounded)": int(round(fwxm_center_idx)),
            "center index (exact)": fwxm_center_idx,
            "center value (@rounded)": float(self._y_original_to_interp(int(round(fwxm_center_idx)))),
            "left index (exact)": left_idx,
            "left index (rounded)": int(round(left_idx)),
            "left value (@rounded)": float(self._y_original_to_interp(int(round(left_idx)))),
            "right index (exact)": right_idx,
            "right index (rounded)": int(round(right_idx)),
            "right value (@rounded)": float(self._y_original_to_interp(int(round(right_idx)))),
            "field values": self._y_original_to_interp(self.x_indices[int(round(left_idx)): int(round(right_idx))]),
            "peak_props": peak_props,
        }
        if self.dpmm:
            data["width (exact) mm"] = data["width (exact)"] / self.dpmm
            data["left distance (exact) mm"] = abs(data["center index (exact)"] - data["left index (exact)"]) / self.dpmm
            data["right distance (exact) mm"] = abs(data["right index (exact)"] - data["center index (exact)"]) / self.dpmm
        return data

    def _sample_points_in_physical_window(self, left_edge: float, right_edge: float):
        """profile.py:1237-1283."""
        xi = self.x_indices
        lo, hi = (left_edge, right_edge) if left_edge <= right_edge else (right_edge, left_edge)

        def nearest(position) -> int:
            return int(np.abs(xi - position).argmin())

        # three tiers, each tried only while fewer than three samples are in hand: every sample inside the closed window;
        # from the sample nearest one end to the sample nearest the other; three samples about the one nearest the middle
        take = slice(int(np.searchsorted(xi, lo, side="left")), int(np.searchsorted(xi, hi, side="right")))
        if take.stop - take.start < 3:
            first, last = sorted((nearest(lo), nearest(hi)))
            take = slice(first, last + 1)
        if take.stop - take.start < 3:
            end = min(len(xi), max(0, nearest((lo + hi) / 2) - 1) + 3)
            take = slice(max(0, end - 3), end)
        xs = xi[take]
        return xs, self._y_original_to_interp(xs)

    def field_data(self, in_field_ratio: float = 0.8, slope_exclusion_ratio=0.2) -> dict:
        """profile.py:1463-1633: in-field window, two edge-slope regressions, quadratic "top"."""
        if not 0 <= in_field_ratio <= 1.0 or not 0 <= slope_exclusion_ratio <= 1.0:
            raise ValueError("in_field_ratio and slope_exclusion_ratio must be within (0, 1)")
        if slope_exclusion_ratio >= in_field_ratio:
            raise ValueError("The exclusion region must be smaller than the field ratio")
        # which centre the windows hang on, and the full field width they are fractions of
        if self._edge_method == Edge.FWHM:
            half_max = self.fwxm_data(x=50)
            beam, span = half_max["center index (exact)"], half_max["width (exact)"]
        else:
            edges = self.inflection_data()
            beam, span = self.beam_center()["index (exact)"], edges["right index (exact)"] - edges["left index (exact)"]
        cax = self.geometric_center()["index (exact)"]
        anchor = cax if self._centering == Centering.GEOMETRIC_CENTER else beam
        field = _Span.about(anchor, in_field_ratio * span)                    # the in-field window
        core = _Span.about(anchor, slope_exclusion_ratio * field.width)       # its flat core, left out of the slope fits
        window = self._sample_points_in_physical_window
        slope = {"left": _linregress(*window(field.lo, core.lo)), "right": _linregress(*window(core.hi, field.hi))}
        top_x, top_y = window(core.lo, core.hi)
        parabola = np.polyfit(top_x, top_y, deg=2)
        top_at, top_value = _bounded_top(parabola, top_x[0] + abs(top_x[-1] - top_x[0]) / 2, top_x[0], top_x[-1])
        # the in-field samples: the index grid moved by the centre's fractional part, so that both halves hold the same
        # number of points (the reference's RAM-4559 note)
        grid = self.x_indices + (anchor - int(round(anchor)))
        first, last = (int(np.abs(grid - edge).argmin()) for edge in (field.lo, field.hi))
        at = self._y_original_to_interp
        out = {"width (exact)": field.width}
        out |= _index_entries("beam center", beam, value=at(round(beam)))
        out |= _index_entries("cax", cax, value=at(round(cax)))
        out |= _index_entries("left", field.lo, value=at(round(field.lo)))
        for side in ("left", "right"):
            out[f"{side} slope"], out[f"{side} intercept"] = slope[side]
        out |= _index_entries("left inner", core.lo) | _index_entries("right inner", core.hi)
        out |= _index_entries('"top"', top_at) | {'"top" value (@exact)': top_value, "top params": parabola}
        out |= _index_entries("right", field.hi, value=at(round(field.hi)))
        out["field values"] = at(grid[first:last + 1])
        if self.dpmm:
            d = self.dpmm
            out["width (exact) mm"] = out["width (exact)"] / d
            out["left slope (%/mm)"] = out["left slope"] * d * 100
            out["right slope (%/mm)"] = out["right slope"] * d * 100
            out["left distance->beam center (exact) mm"] = abs(out["beam center index (exact)"] - out["left index (exact)"]) / d
            out["right distance->beam center (exact) mm"] = abs(out["right index (exact)"] - out["beam center index (exact)"]) / d
            out["left distance->CAX (exact) mm"] = abs(out["cax index (exact)"] - out["left index (exact)"]) / d
            out["right distance->CAX (exact) mm"] = abs(out["cax index (exact)"] - out["right index (exact)"]) / d
            out["left distance->top (exact) mm"] = abs(out['"top" index (exact)'] - out["left index (exact)"]) / d
            out["right distance->top (exact) mm"] = abs(out['"top" index (exact)'] - out["right index (exact)"]) / d
            out['"top"->beam center (exact) mm'] = (out['"top" index (exact)'] - out["beam center index (exact)"]) / d
            out['"top"->CAX (exact) mm'] = abs(out['"top" index (exact)'] - out["cax index (exact)"]) / d
        return out

    def field_calculation(self, in_field_ratio: float = 0.8, calculation: str = "mean",
                          slope_exclusion_ratio: float = 0.2):
        """profile.py:1910-1937."""
        fv = self.field_data(in_field_ratio, slope_exclusion_ratio=slope_exclusion_ratio)["field values"]
        if calculation == "mean":
            return fv.mean()
        if calculation == "median":
            return float(np.median(fv))
        if calculation == "max":
            return fv.max()
        if calculation == "min":
            return fv.min()


@dataclass
class HillEdgesBatch:
    """What ``SingleProfile(..., edge_detection_method=INFLECTION_HILL)`` holds and ``inflection_data()`` returns, for every
    row of a batch (:func:`single_profile_hill_batch`); everything stays on the device."""

    values: torch.Tensor            # float64 [N, S]   the processed profiles (SingleProfile.values)
    x_indices: np.ndarray           # float64 [S]      SingleProfile.x_indices (shared)
    dpmm: float | None
    params: torch.Tensor            # float64 [N, 2, 4] "left / right Hill params"
    index: torch.Tensor             # float64 [N, 2]   "left / right index (exact)": the Hill inflection points
    value: torch.Tensor             # float64 [N, 2]   "left / right value (@exact)"
    derivative_edges: torch.Tensor  # float64 [N, 2]   the windows' centres (the INFLECTION_DERIVATIVE edges)
    info: torch.Tensor              # int32 [N, 2]     MINPACK info: 1-4 = converged (what curve_fit accepts); 5-8 = curve_fit
                                    #                  raises RuntimeError; -1 = fewer than four samples in the window (TypeError);
                                    #                  -2 = no derivative peak / valley (IndexError); -3 = more extrema than peak_cap;
                                    #                  -4 = NaN / infinity in a window (curve_fit's check_finite: ValueError)
    nfev: torch.Tensor              # int32 [N, 2]
    last_step: torch.Tensor | None = None   # float64 [N, 2]  length of the last accepted Levenberg-Marquardt step relative to the
                                    #                  parameter vector (MINPACK's scaled variables): see ``settled``

    SETTLED_STEP = 1.0e-6

    @property
    def settled(self) -> torch.Tensor:
        """bool [N, 2]: the fit converged AND its last accepted step was below 1e-6 of the parameter vector.  MINPACK stops on
        the reduction of the sum of squares; in a flat valley that happens while the parameters still move, and where exactly
        depends on the last bit of ``pow`` -- scipy, numpy's vectorised ``pow`` and this device each stop at a slightly
        different point of such a valley.  Settled fits reproduce scipy's inflection point to 1e-5 (checked on every window of
        the test sets); the others (a few percent of noisy synthetic windows, none of the reference's own profiles) to ~1e-3."""
        ok = (self.info >= 1) & (self.info <= 4)
        return ok if self.last_step is None else ok & (self.last_step <= self.SETTLED_STEP)

    def inflection_data(self, i: int) -> dict:
        """The reference's dictionary (profile.py:1701-1721) for profile ``i`` -- a host copy of eight numbers."""
        info = self.info[i].cpu().numpy()
        for side, code in zip(("left", "right"), info):
            if code == -3:
                raise _lib_error(f"profile {i}: more derivative extrema than peak_cap; raise it")
            if code == -2:
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            if code == -1:
                raise TypeError("The number of func parameters=4 must not exceed the number of data points")
            if code == -4:
                raise ValueError("array must not contain infs or NaNs")
            if not 1 <= code <= 4:
                raise RuntimeError(f"Optimal parameters not found ({side} penumbra of profile {i}: MINPACK info {code})")
        idx, val, prm = self.index[i].cpu().numpy(), self.value[i].cpu().numpy(), self.params[i].cpu().numpy()
        return {
            "left index (rounded)": int(round(float(idx[0]))),
            "left index (exact)": float(idx[0]),
            "right index (rounded)": int(round(float(idx[1]))),
            "right index (exact)": float(idx[1]),
            "left value (@exact)": float(val[0]),
            "right value (@exact)": float(val[1]),
            "left Hill params": prm[0],
            "right Hill params": prm[1],
        }


    def penumbra(self, lower: int = 20, upper: int = 80) -> dict:
        """``SingleProfile.penumbra`` for the Hill method (profile.py:1852-1908) for every row: the reference's keys, each a
        float64 [N] tensor on the device ("left values" / "right values" are ragged slices of ``values`` between the rounded
        indices and are left to the caller).  Rows whose fit failed hold NaN."""
        if lower > upper:
            raise ValueError("Upper penumbra value must be larger than the lower penumbra value")
        rec = ops.hill_penumbra(self.params, torch.stack([self.index, self.value], dim=-1), lower, upper)   # [N, 2, 6]
        data = {}
        for s, side in enumerate(("left", "right")):
            data[f"{side} {lower}% index (exact)"] = rec[:, s, 0]
            data[f"{side} {lower}% value (exact)"] = rec[:, s, 1]
            data[f"{side} {upper}% index (exact)"] = rec[:, s, 2]
            data[f"{side} {upper}% value (exact)"] = rec[:, s, 3]
            data[f"{side} penumbra width (exact)"] = rec[:, s, 4]
            data[f"{side} gradient (exact)"] = rec[:, s, 5]
            if self.dpmm:
                data[f"{side} gradient (exact) %/mm"] = rec[:, s, 5] * self.dpmm * 100
                data[f"{side} penumbra width (exact) mm"] = rec[:, s, 4] / self.dpmm
        return data


def _lib_error(msg: str):
    from ._lib import PylinacHipError

    return PylinacHipError(msg)


def _hill_edges_stage(vals: torch.Tensor, xi_dev: torch.Tensor, span: float, edge_smoothing_ratio: float, ratio: float,
                      peak_cap: int):
    """``SingleProfile.inflection_data`` (profile.py:1635-1721) for every row: seven launches, no host round trip."""
    n, s = vals.shape
    sm = ops.gaussian_filter1d(vals, edge_smoothing_ratio * s)
    d1 = ops.gradient1d(sm)
    cap = max(min(peak_cap, s // 2 + 1), 1)
    pk = ops.find_peaks_batch(d1, cap=cap, threshold=0.8, peak_separation=0.05)      # MultiProfile(d1).find_peaks
    vl = ops.find_peaks_batch(-d1, cap=cap, threshold=0.8, peak_separation=0.05)     # MultiProfile(d1).find_valleys
    mmax = max(int(ratio * span) + 3, 4)
    if mmax > 1024:
        raise ValueError("hill_window_ratio x field width: more than 1024 samples per window")
    xw, yw, lens, edges = ops.hill_windows(xi_dev, vals, pk, vl, ratio, mmax)
    params, info, nfev, step = ops.hill_fit(xw, yw, lens, last_step=True)
    infl = ops.hill_inflection(params)
    no_edge = torch.isnan(edges).any(dim=1).repeat_interleave(2)
    overflow = ((pk.status != 0) | (vl.status != 0)).repeat_interleave(2)
    info = torch.where(no_edge, torch.full_like(info, -2), info)
    info = torch.where(overflow, torch.full_like(info, -3), info)
    return params.view(n, 2, 4), infl.view(n, 2, 2), edges, info.view(n, 2), nfev.view(n, 2), step.view(n, 2)


def _batch_constructor(values, dpmm, interpolation, ground, interpolation_resolution_mm, interpolation_factor):
    """``SingleProfile.__init__`` up to the normalisation (pylinac/core/profile.py:1182-1205) for every row of ``values`` [N, L]
    (x = 0 .. L-1): resampling and grounding -> (fitted float64 [N, S] on the device, x_indices numpy [S], x_indices on the
    device)."""
    interp = _enum(interpolation, Interpolation)
    v = values if isinstance(values, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64))
    if v.dim() != 2:
        raise ValueError("values must be [N, L]")
    if v.numel() == 0:
        raise ValueError("Array must not be empty")
    if not v.is_cuda:
        v = v.to(au._device())
    v = v.to(torch.float64).contiguous()
    n, length = v.shape
    x_values = np.arange(length)
    if interp == Interpolation.NONE:
        fitted, x_indices = v, x_values.astype(np.float64)
    else:                                                   # SingleProfile._interpolate (profile.py:1306-1360)
        if dpmm is not None:
            samples = int(round(length / (dpmm * interpolation_resolution_mm)))
        else:
            samples = int(round(length * interpolation_factor))
        offset = 0.5 - 1 / (2 * (samples / length))
        x_indices = np.linspace(x_values[0] - offset, x_values[-1] + offset, num=samples)
        xs = torch.from_numpy(x_values.astype(np.float64)).to(v.device)
        fitted = ops.interp1d(xs, v, torch.from_numpy(x_indices).to(v.device),
                              kind="linear" if interp == Interpolation.LINEAR else "cubic")
    xi_dev = torch.from_numpy(np.ascontiguousarray(x_indices)).to(v.device)
    if ground:
        fitted = ops.ground(fitted.unsqueeze(1)).squeeze(1)
    return fitted.contiguous(), x_indices, xi_dev


def _batch_normalize(fitted: torch.Tensor, norm: Normalization, beam_center_value):
    """``SingleProfile._normalize`` (profile.py:1362-1371) per row; ``beam_center_value(fitted)`` -> float64 [N] is the edge
    method's ``beam_center()["value (@rounded)"]`` on the unnormalised profiles."""
    if norm == Normalization.MAX:
        return ops.normalize(fitted.unsqueeze(1)).squeeze(1).contiguous()
    if norm == Normalization.GEOMETRIC_CENTER:              # array_utils.geometric_center_value
        s = fitted.shape[1]
        gc = (fitted[:, s // 2] + fitted[:, s // 2 - 1]) / 2.0 if s % 2 == 0 else fitted[:, (s - 1) // 2]
        return ops.normalize(fitted.unsqueeze(1), gc.contiguous()).squeeze(1).contiguous()
    if norm == Normalization.BEAM_CENTER:
        return ops.normalize(fitted.unsqueeze(1), beam_center_value(fitted).contiguous()).squeeze(1).contiguous()
    return fitted


def single_profile_hill_batch(values, dpmm: float | None = None, interpolation=Interpolation.LINEAR, ground: bool = True,
                              interpolation_resolution_mm: float = 0.1, interpolation_factor: float = 10,
                              normalization_method=Normalization.BEAM_CENTER, edge_smoothing_ratio: float = 0.003,
                              hill_window_ratio: float = 0.1, peak_cap: int = 32) -> HillEdgesBatch:
    """``SingleProfile(values_i, ..., edge_detection_method=Edge.INFLECTION_HILL)`` followed by ``inflection_data()`` for every
    row of ``values`` [N, L] (equal lengths, x = 0 .. L-1: the rows of an image, a detector array's frames) with no per-profile
    host call: the constructor's resampling, grounding and normalisation (pylinac/core/profile.py:1165-1215), the smoothed
    derivative and its extrema, both penumbra windows and both four-parameter fits (profile.py:1635-1721, hill.py:18-36) are
    batched launches.  BEAM_CENTER normalisation runs the edge search twice, like the reference's constructor (the norm value is
    the profile at the midpoint of the two Hill inflection points).  Rows whose search or fit fails are reported in ``info``
    (their results are NaN), not raised."""
    norm = _enum(normalization_method, Normalization)
    fitted, x_indices, xi_dev = _batch_constructor(values, dpmm, interpolation, ground, interpolation_resolution_mm,
                                                   interpolation_factor)
    span = float(x_indices[-1] - x_indices[0])

    first_info = []

    def beam_center_value(unnormalised):                    # beam_center() (profile.py:1390-1409) on the unnormalised profile
        _, infl, _, info0, _, _ = _hill_edges_stage(unnormalised, xi_dev, span, edge_smoothing_ratio, hill_window_ratio, peak_cap)
        first_info.append(info0)
        left, right = infl[:, 0, 0], infl[:, 1, 0]
        mid = torch.round(left + (right - left) / 2)        # int(round(mid_point)): half to even, like python's
        return ops.profile_lookup(xi_dev, unnormalised, mid.contiguous())

    fitted = _batch_normalize(fitted, norm, beam_center_value)
    params, infl, edges, info, nfev, step = _hill_edges_stage(fitted, xi_dev, span, edge_smoothing_ratio, hill_window_ratio, peak_cap)
    if first_info:
        # a fit the constructor's own beam_center() could not make (MINPACK info outside 1..4: curve_fit raises RuntimeError
        # in the reference's __init__) normalised this row by a value at a non-converged midpoint: the row keeps THAT code
        bad = (first_info[0] < 1) | (first_info[0] > 4)
        info = torch.where(bad, first_info[0], info)
    return HillEdgesBatch(values=fitted, x_indices=x_indices, dpmm=dpmm,
                          params=params, index=infl[..., 0], value=infl[..., 1], derivative_edges=edges, info=info, nfev=nfev,
                          last_step=step)


@dataclass
class FWXMEdgesBatch:
    """What ``SingleProfile(..., edge_detection_method=FWHM)`` holds, for every row of a batch (:func:`single_profile_fwhm_batch`);
    ``fwxm_data(x)`` returns the reference's keys as float64 [N] tensors on the device."""

    values: torch.Tensor            # float64 [N, S]   the processed profiles (SingleProfile.values)
    x_indices: np.ndarray           # float64 [S]
    dpmm: float | None

    def _edges(self, x: float, values: torch.Tensor | None = None):
        """left / right FWXM edges in ORIGINAL coordinates and the peak count per row (0: the reference raises IndexError)"""
        v = self.values if values is None else values
        res = ops.find_peaks_batch(v, cap=1, fwxm_height=x / 100, max_number=1)
        s = v.shape[1]
        dev = v.device
        xi = torch.from_numpy(np.ascontiguousarray(self.x_indices)).to(dev)
        ips = torch.stack([res.props[:, 4, 0], res.props[:, 5, 0]], dim=0).reshape(1, -1)      # left_ips, right_ips
        ips = torch.where(res.count.repeat(2).reshape(1, -1) > 0, ips, torch.zeros_like(ips))
        orig = ops.index_to_original(xi, ips.contiguous()).reshape(2, -1)                      # _x_interp_to_original
        nan = torch.full_like(orig[0], float("nan"))
        ok = res.count > 0
        return torch.where(ok, orig[0], nan), torch.where(ok, orig[1], nan), res.count, xi

    def fwxm_data(self, x: int = 50) -> dict:
        """``SingleProfile.fwxm_data`` (profile.py:1411-1461) for every row; rows without a peak hold NaN ("peaks" is 0 there).
        "field values" (a ragged slice) and "peak_props" are left out."""
        if not 0 <= x <= 100:
            raise ValueError("x must be within (0, 100)")
        left, right, count, xi = self._edges(x)
        width = right - left
        centre = (right - left) / 2 + left
        at = lambda idx: ops.profile_lookup(xi, self.values, torch.nan_to_num(torch.round(idx)).contiguous())
        bad = count <= 0
        mask = lambda t: torch.where(bad, torch.full_like(t, float("nan")), t)
        data = {
            "peaks": count,
            "width (exact)": width,
            "width (rounded)": torch.round(width),
            "center index (rounded)": torch.round(centre),
            "center index (exact)": centre,
            "center value (@rounded)": mask(at(centre)),
            "left index (exact)": left,
            "left index (rounded)": torch.round(left),
            "left value (@rounded)": mask(at(left)),
            "right index (exact)": right,
            "right index (rounded)": torch.round(right),
            "right value (@rounded)": mask(at(right)),
        }
        if self.dpmm:
            data["width (exact) mm"] = width / self.dpmm
            data["left distance (exact) mm"] = torch.abs(centre - left) / self.dpmm
            data["right distance (exact) mm"] = torch.abs(right - centre) / self.dpmm
        return data

    def penumbra(self, lower: int = 20, upper: int = 80) -> dict:
        """``SingleProfile.penumbra`` for the FWHM method (profile.py:1759-1790) for every row: the FWXM edges at ``lower`` and
        ``upper`` per cent ("left values" / "right values", ragged slices, are left out)."""
        if lower > upper:
            raise ValueError("Upper penumbra value must be larger than the lower penumbra value")
        lo, up = self.fwxm_data(lower), self.fwxm_data(upper)
        data = {}
        for side in ("left", "right"):
            data[f"{side} {lower}% index (exact)"] = lo[f"{side} index (exact)"]
            data[f"{side} {lower}% value (@rounded)"] = lo[f"{side} value (@rounded)"]
            data[f"{side} {upper}% index (exact)"] = up[f"{side} index (exact)"]
            data[f"{side} {upper}% value (@rounded)"] = up[f"{side} value (@rounded)"]
            data[f"{side} penumbra width (exact)"] = torch.abs(up[f"{side} index (exact)"] - lo[f"{side} index (exact)"])
            if self.dpmm:
                data[f"{side} penumbra width (exact) mm"] = data[f"{side} penumbra width (exact)"] / self.dpmm
        return data

    def beam_center(self) -> dict:
        """``SingleProfile.beam_center`` for the FWHM method (profile.py:1390-1399)."""
        d = self.fwxm_data(50)
        return {"index (rounded)": d["center index (rounded)"], "index (exact)": d["center index (exact)"],
                "value (@rounded)": d["center value (@rounded)"]}


def single_profile_fwhm_batch(values, dpmm: float | None = None, interpolation=Interpolation.LINEAR, ground: bool = True,
                              interpolation_resolution_mm: float = 0.1, interpolation_factor: float = 10,
                              normalization_method=Normalization.BEAM_CENTER) -> FWXMEdgesBatch:
    """``SingleProfile(values_i, ...)`` with the default FWHM edge method for every row of ``values`` [N, L] (equal lengths) in
    batched launches: resampling, grounding, the beam-centre (or maximum / geometric-centre) normalisation
    (pylinac/core/profile.py:1165-1215, 1362-1409); ``fwxm_data(x)`` / ``beam_center()`` of the result answer for all rows."""
    norm = _enum(normalization_method, Normalization)
    fitted, x_indices, _ = _batch_constructor(values, dpmm, interpolation, ground, interpolation_resolution_mm,
                                              interpolation_factor)
    fitted = _batch_normalize(fitted, norm, lambda unnormalised: FWXMEdgesBatch(unnormalised, x_indices, dpmm)
                              .fwxm_data(50)["center value (@rounded)"])
    return FWXMEdgesBatch(values=fitted, x_indices=x_indices, dpmm=dpmm)


@dataclass
class InflectionEdgesBatch:
    """``SingleProfile(..., edge_detection_method=INFLECTION_DERIVATIVE)`` and its ``inflection_data()`` (profile.py:1635-1674,
    1711-1721) for every row of a batch (:func:`single_profile_inflection_batch`): float64 [N] tensors under the reference's
    keys; rows whose derivative has no peak or valley (the reference raises IndexError) hold NaN."""

    values: torch.Tensor            # float64 [N, S]
    x_indices: np.ndarray           # float64 [S]
    dpmm: float | None
    edges: torch.Tensor             # float64 [N, 2]   left / right index (exact)
    status: torch.Tensor            # int32 [N]        0 ok, 2 no derivative peak / valley, 3 more extrema than peak_cap

    def inflection_data(self) -> dict:
        xi = torch.from_numpy(np.ascontiguousarray(self.x_indices)).to(self.values.device)
        left, right = self.edges[:, 0].contiguous(), self.edges[:, 1].contiguous()
        at = lambda q: torch.where(torch.isnan(q), q, ops.profile_lookup(xi, self.values, torch.nan_to_num(q).contiguous()))
        return {
            "left index (rounded)": torch.round(left),
            "left index (exact)": left,
            "right index (rounded)": torch.round(right),
            "right index (exact)": right,
            "left value (@rounded)": at(torch.round(left)),
            "left value (@exact)": at(left),
            "right value (@rounded)": at(torch.round(right)),
            "right value (@exact)": at(right),
        }


def _inflection_edges_stage(vals: torch.Tensor, xi_dev: torch.Tensor, edge_smoothing_ratio: float, peak_cap: int):
    """the outermost extrema of the smoothed derivative in original coordinates (profile.py:1660-1674) -> ([N, 2], status)"""
    n, s = vals.shape
    sm = ops.gaussian_filter1d(vals, edge_smoothing_ratio * s)
    d1 = ops.gradient1d(sm)
    cap = max(min(peak_cap, s // 2 + 1), 1)
    pk = ops.find_peaks_batch(d1, cap=cap, threshold=0.8, peak_separation=0.05)
    vl = ops.find_peaks_batch(-d1, cap=cap, threshold=0.8, peak_separation=0.05)
    _, _, _, edges = ops.hill_windows(xi_dev, vals, pk, vl, 0.0, 4)     # (the windows themselves are not needed here)
    status = torch.where(torch.isnan(edges).any(dim=1), 2, 0).to(torch.int32)
    status = torch.where((pk.status != 0) | (vl.status != 0), torch.full_like(status, 3), status)
    return edges, status


def single_profile_inflection_batch(values, dpmm: float | None = None, interpolation=Interpolation.LINEAR, ground: bool = True,
                                    interpolation_resolution_mm: float = 0.1, interpolation_factor: float = 10,
                                    normalization_method=Normalization.BEAM_CENTER, edge_smoothing_ratio: float = 0.003,
                                    peak_cap: int = 32) -> InflectionEdgesBatch:
    """``SingleProfile(values_i, ..., edge_detection_method=Edge.INFLECTION_DERIVATIVE)`` for every row of ``values`` [N, L] in
    batched launches (the constructor of :func:`single_profile_hill_batch` without the fits: the beam centre is the midpoint of
    the derivative's outermost extrema, profile.py:1400-1409)."""
    norm = _enum(normalization_method, Normalization)
    fitted, x_indices, xi_dev = _batch_constructor(values, dpmm, interpolation, ground, interpolation_resolution_mm,
                                                   interpolation_factor)

    def beam_center_value(unnormalised):
        e, _ = _inflection_edges_stage(unnormalised, xi_dev, edge_smoothing_ratio, peak_cap)
        mid = torch.round(e[:, 0] + (e[:, 1] - e[:, 0]) / 2)
        return torch.where(torch.isnan(mid), mid, ops.profile_lookup(xi_dev, unnormalised, torch.nan_to_num(mid).contiguous()))

    fitted = _batch_normalize(fitted, norm, beam_center_value)
    edges, status = _inflection_edges_stage(fitted, xi_dev, edge_smoothing_ratio, peak_cap)
    return InflectionEdgesBatch(values=fitted, x_indices=x_indices, dpmm=dpmm, edges=edges, status=status)


def _linregress(x, y):
    """slope / intercept of ``scipy.stats.linregress`` (ssxym / ssxm from the biased covariance matrix)."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    xmean, ymean = np.mean(x), np.mean(y)
    ssxm, ssxym, _, _ = np.cov(x, y, bias=1).flat
    slope = ssxym / ssxm
    return slope, ymean - slope * xmean


class _Span:
    """An index interval hung symmetrically on a centre: ``about(c, full)`` -> [c - full / 2, c + full / 2]; ``width`` is the
    difference of the two ROUNDED ends (what the reference carries forward), not ``full``."""

    __slots__ = ("lo", "hi")

    def __init__(self, lo: float, hi: float):
        self.lo, self.hi = lo, hi

    @classmethod
    def about(cls, centre: float, full: float) -> "_Span":
        return cls(centre - full / 2, centre + full / 2)

    @property
    def width(self) -> float:
        return self.hi - self.lo


def _index_entries(name: str, index: float, value=None) -> dict:
    """The reference's result-dictionary triplet for one position: ``<name> index (exact)``, ``... (rounded)`` and -- where it
    reports one -- ``<name> value (@rounded)``."""
    entries = {f"{name} index (exact)": index, f"{name} index (rounded)": int(round(index))}
    if value is not None:
        entries[f"{name} value (@rounded)"] = value
    return entries


def _bounded_top(fit_params, x0: float, lo: float, hi: float):
    """The reference maximises the fitted parabola with ``scipy.optimize.minimize(-p, x0, bounds)`` (L-BFGS-B,
    profile.py:1537-1548) and reports wherever that stops -- on a flat top it can stop well short of the
    vertex -- so the same routine is called here: one scalar problem per profile, on the host like the
    reference."""
    from scipy.optimize import minimize

    def poly_func(x):
        return -(fit_params[0] * (x**2) + fit_params[1] * x + fit_params[2])

    min_f = minimize(poly_func, x0=(x0,), bounds=((lo, hi),))
    return min_f.x[0], -min_f.fun
