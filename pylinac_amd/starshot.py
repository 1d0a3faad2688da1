"""Starshot per-image measurement on the device (the Starshot caller of SURVEY.md section 8 rows a5-a12 that
BASELINE.json's north_star names; call stack in SURVEY.md section 3.3).

Mirrors ``pylinac.starshot.Starshot.analyze`` (pylinac/starshot.py:229-401) for an array image: histogram inversion
check and grounding, the automatic start point (maxima of the central third along both axes -> FW80M centres, 90th
percentile), the ``StarProfile`` (a collapsed circle profile of 20 radii at 3x sampling, rolled to its minimum,
Gaussian-filtered, grounded, FWXM peak search), peak pairing into radiation lines and the wobble circle.

Dense work -- percentiles, grounding / inversion, axis maxima, the 20-radius nearest-neighbour gather, the 1-D Gaussian
and the peak / FWXM search -- runs in the kernels behind ``ArrayImage`` / ``CollapsedCircleProfile`` / ``FWXMProfile``.
The wobble fit is the reference's own per-dataset optimiser (Nelder-Mead over <= a dozen lines, ``scipy.optimize``) on
the host, and so is the parameter sweep that retries with other radii / peak heights.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from itertools import product

import numpy as np
import torch
from scipy import optimize

from . import ops
from .array_utils import _Staged, _device
from .geometry import Circle
from .image import ArrayImage
from .profile import CollapsedCircleProfile, FWXMProfile, MultiProfile, Point, _linear_at


def _point(p) -> Point:
    if isinstance(p, Point):
        return p
    if hasattr(p, "x"):
        return Point(x=p.x, y=p.y)
    return Point(x=p[0], y=p[1])


def _distance(a: Point, b: Point) -> float:
    """Point.distance_to (pylinac/core/geometry.py:122-140)"""
    return math.sqrt((a.x - b.x) ** 2 + (a.y - b.y) ** 2 + (a.z - b.z) ** 2)


class Line:
    """pylinac/core/geometry.py:493-584: the two methods the star-shot analysis uses."""

    def __init__(self, point1, point2):
        self.point1, self.point2 = _point(point1), _point(point2)

    @property
    def m(self) -> float:
        """slope; vertical lines: the reference catches ZeroDivisionError (python floats) or divides numpy scalars"""
        dy, dx = self.point1.y - self.point2.y, self.point1.x - self.point2.x
        try:
            return dy / dx
        except ZeroDivisionError:
            return float("inf")

    def distance_to(self, point) -> float:
        p = _point(point)
        pt = np.array([p.x, p.y, p.z], dtype=float)
        lp1 = np.array([self.point1.x, self.point1.y, self.point1.z], dtype=float)
        lp2 = np.array([self.point2.x, self.point2.y, self.point2.z], dtype=float)
        numerator = np.sqrt(np.sum(np.power(np.cross((lp2 - lp1), (lp1 - pt)), 2)))
        denominator = np.sqrt(np.sum(np.power(lp2 - lp1, 2)))
        return numerator / denominator


class Wobble:
    """starshot.py:683-698"""

    def __init__(self):
        self.center = Point()
        self.radius = None
        self.radius_mm = 0

    @property
    def diameter_mm(self) -> float:
        return self.radius_mm * 2


class LineManager:
    """starshot.py:701-762"""

    def __init__(self, points, focus_point: Point, dpmm: float):
        self.focus_point, self.dpmm = focus_point, dpmm
        num_rad_lines = int(len(points) / 2)
        self.lines = [Line(points[k], points[k + num_rad_lines]) for k in range(num_rad_lines)]
        for line in self.lines:
            if line.distance_to(focus_point) > 10 * dpmm:
                raise ValueError("The radiation lines are not near the center of the image. "
                                 "This could be due to missing spoke halves, such as in a gantry starshot.")

    def __getitem__(self, item):
        return self.lines[item]

    def __len__(self):
        return len(self.lines)


class StarProfile(CollapsedCircleProfile):
    """starshot.py:765-814"""

    def __init__(self, image: ArrayImage, start_point, radius: float, min_peak_height: float, fwhm: bool):
        start_point = _point(start_point)
        rows, cols = image.shape[:2]
        dist2edge_min = min(rows - start_point.y, cols - start_point.x, start_point.y, start_point.x)   # image.py:817-837
        super().__init__(center=start_point, radius=dist2edge_min * radius, image_array=image.array, width_ratio=0.1,
                         sampling_ratio=3)
        self.get_peaks(min_peak_height, fwhm=fwhm)

    @classmethod
    def _ring(cls, shape, start_point, radius: float) -> "StarProfile":
        """The same ring WITHOUT its samples (``analyze_batch`` gathers the rings of many frames in one launch and hands
        each profile its ``values``): centre, radius, the array-size check, 20 radii at +-10 %, sampling ratio 3."""
        self = object.__new__(cls)
        start_point = _point(start_point)
        rows, cols = shape
        dist2edge_min = min(rows - start_point.y, cols - start_point.x, start_point.y, start_point.x)
        Circle.__init__(self, start_point, dist2edge_min * radius)
        if cols < self.radius + self.center.x or rows < self.radius + self.center.y:        # CircleProfile._ensure_array_size
            raise ValueError("Array size not large enough to compute profile")
        self.image_array = None
        self.start_angle, self.ccw, self.sampling_ratio = 0, True, 3
        self.width_ratio, self.num_profiles = 0.1, 20
        self._x_locations = self._y_locations = None
        MultiProfile.__init__(self, None)
        return self

    def get_peaks(self, min_peak_height, min_peak_distance=0.02, fwhm: bool = True):
        roll_amount = np.where(self.values == self.values.min())[0][0]
        self.roll(roll_amount)
        self.filter(size=0.003, kind="gaussian")
        self.ground()
        if fwhm:
            self.find_fwxm_peaks(threshold=min_peak_height, min_distance=min_peak_distance)
        else:
            self.find_peaks(min_peak_height, min_peak_distance)


_NO_LINES = ("The algorithm was unable to properly detect the radiation lines. Try setting "
             "recursive to True or lower the minimum peak height")
_NO_WOBBLE = ("The algorithm was unable to determine a reasonable wobble. Try setting "
              "recursive to False and manually adjusting algorithm parameters")


def _retry_sweep(radius: float, min_peak_height: float):
    """starshot.py:320-327: the (radius, peak height) pairs tried after the caller's own, in the reference's order."""
    peak_heights = np.append(min_peak_height, np.linspace(0.05, 0.95, 10))
    radii = np.append(radius, np.linspace(0.95, 0.1, 10))
    return product(radii, peak_heights)


def calculate_angles(lines) -> list:
    """starshot.py:817-834"""
    angles = []
    for line in lines:
        phi_deg = math.degrees(math.atan(line.m)) - 90
        if phi_deg > 90:
            phi_deg -= 180
        elif phi_deg <= -90:
            phi_deg += 180
        angles.append(phi_deg)
    return angles


class Starshot:
    """``pylinac.starshot.Starshot`` for an array image (the reference loads files; ``Starshot(array, dpi=, sid=)``
    here).  ``analyze`` has the reference's arguments, retry sweep and error messages; results: ``wobble`` (centre,
    ``radius`` px, ``radius_mm``, ``diameter_mm``), ``lines``, ``angles``, ``circle_profile``, ``passed``."""

    def __init__(self, array, dpi: float | None = None, sid: float | None = None):
        self.image = array if isinstance(array, ArrayImage) else ArrayImage(array, dpi=dpi, sid=sid)
        self.wobble = Wobble()
        self.tolerance = 1
        if self.image.dpmm is None:
            raise ValueError("DPI was not a tag in the image nor was it passed in. Please pass a DPI value")
        if self.image.sid is None:
            raise ValueError("Source-to-Image distance was not an image tag and was not passed in. Please pass an SID value.")

    def _get_reasonable_start_point(self):
        """starshot.py:197-227: FW80M centres of the axis maxima of the central third, and its 90th percentile."""
        a = self.image.array
        top_third = int(a.shape[0] / 3)
        bottom_third = int(top_third * 2)
        left_third = int(a.shape[1] / 3)
        right_third = int(left_third * 2)
        central = np.ascontiguousarray(a[top_third:bottom_third, left_third:right_third])
        staged = _Staged(central).t
        x_max = ops.reduce_axis(staged, 0, "max")[0].cpu().numpy().astype(a.dtype)       # np.max(central, 0)
        y_max = ops.reduce_axis(staged, 1, "max")[0].cpu().numpy().astype(a.dtype)       # np.max(central, 1)
        fwxm_x_point = round(FWXMProfile(values=x_max, fwxm_height=80).center_idx) + left_third
        fwxm_y_point = round(FWXMProfile(values=y_max, fwxm_height=80).center_idx) + top_third
        if central.dtype in (np.uint16, np.int16):
            local_max = float(ops.percentile(staged, [90])[0][0])
        else:
            from .canny import _percentile_f64

            local_max = float(_percentile_f64(_Staged(central.astype(np.float64, copy=False)).t, 90)[0])
        return Point(x=fwxm_x_point, y=fwxm_y_point), local_max

    def analyze(self, radius: float = 0.85, min_peak_height: float = 0.25, max_wobble_diameter: float = 2.0,
                tolerance: float = 1.0, start_point=None, fwhm: bool = True, recursive: bool = True,
                invert: bool = False) -> None:
        """starshot.py:229-304"""
        if not 0.2 <= radius <= 0.95:
            raise ValueError("radius must be within (0.2, 0.95)")
        if not 0.05 <= min_peak_height <= 0.95:
            raise ValueError("min_peak_height must be within (0.05, 0.95)")
        self.tolerance = tolerance
        self.image.check_inversion_by_histogram(percentiles=[4, 50, 96])
        self.image.ground()
        if invert:
            self.image.invert()
        auto_point, local_max = self._get_reasonable_start_point()
        if start_point is None:
            start_point = auto_point
        self._get_reasonable_wobble(_point(start_point), fwhm, min_peak_height, radius, recursive, local_max,
                                    max_wobble_diameter)
        self.angles = calculate_angles(self.lines)

    def _get_reasonable_wobble(self, start_point, fwhm, min_peak_height, radius, recursive, local_max,
                               max_wobble_diameter) -> None:
        """starshot.py:306-376"""
        focus_point = copy.copy(start_point)
        radius_and_peak_gen = _retry_sweep(radius, min_peak_height)
        while True:
            try:
                min_height = min_peak_height * local_max
                self.circle_profile = StarProfile(self.image, focus_point, radius, min_height, fwhm)
                if self._accept(focus_point, recursive, max_wobble_diameter):
                    return
            except ValueError:
                pass
            try:
                radius, min_peak_height = next(radius_and_peak_gen)
            except StopIteration:
                raise RuntimeError(_NO_WOBBLE)

    def _accept(self, focus_point, recursive, max_wobble_diameter) -> bool:
        """The loop body of starshot.py:320-376 behind the star profile: an even number (>= 6) of peaks, lines that pass
        the focus point (``LineManager`` raises ValueError otherwise), the wobble fit, and the sanity test on it.  False
        (or a ValueError) sends the caller to the next (radius, peak height) of the sweep."""
        n = len(self.circle_profile.peaks)
        if n < 6 or n % 2 != 0:
            if not recursive:
                raise RuntimeError(_NO_LINES)
            return False
        self.lines = LineManager(self.circle_profile.peaks, focus_point=focus_point, dpmm=self.image.dpmm)
        self._find_wobble_minimize()
        focus_near_center = _distance(self.wobble.center, focus_point) < 10 * self.image.dpmm
        return bool((self.wobble.diameter_mm < max_wobble_diameter and focus_near_center) or not recursive)

    def _find_wobble_minimize(self) -> None:
        """starshot.py:378-401: the smallest circle touching every line, Nelder-Mead from the profile centre."""
        sp = self.circle_profile.center

        def distance(p, lines):
            return max(line.distance_to(Point(x=p[0], y=p[1])) for line in lines)

        # the reference starts from Point.as_array() = (x, y, z): a three-parameter simplex whose third coordinate the
        # objective ignores -- kept, because the simplex path (and so the stopping point) depends on it
        res = optimize.minimize(distance, np.array([sp.x, sp.y, sp.z]), args=(self.lines,),
                                method="Nelder-Mead", options={"fatol": 0.001})
        self.wobble.radius = res.fun
        self.wobble.radius_mm = res.fun / self.image.dpmm
        self.wobble.center = Point(x=res.x[0], y=res.x[1])

    @property
    def passed(self) -> bool:
        return bool(self.wobble.radius_mm * 2 < self.tolerance)


# ---------------------------------------------------------------------------------------------------- the batched form
@dataclass
class StarshotBatch:
    """``analyze_batch``'s records, one row per frame.  ``status``: 0 measured; 1 the (radius, peak height) sweep ran out
    ("unable to determine a reasonable wobble": ``Starshot.analyze`` raises RuntimeError there); 2 ``recursive=False`` and the
    lines were not detected (RuntimeError there too); 3 no FW80M peak in the central third (IndexError there).  Rows with a
    non-zero status hold NaN."""

    status: np.ndarray                 # int32 [N]
    wobble_center: np.ndarray          # float64 [N, 2] (x, y) px
    wobble_radius: np.ndarray          # float64 [N] px
    wobble_radius_mm: np.ndarray
    wobble_diameter_mm: np.ndarray
    passed: np.ndarray                 # bool [N]
    n_lines: np.ndarray                # int32 [N]
    start_point: np.ndarray            # float64 [N, 2]: the automatic start point (x, y)
    local_max: np.ndarray              # float64 [N]: 90th percentile of the central third
    inverted: np.ndarray               # bool [N]: the histogram check inverted the frame
    radius: np.ndarray                 # float64 [N]: the (radius, min_peak_height) pair that was accepted
    min_peak_height: np.ndarray
    analyzers: list = field(default_factory=list)   # per frame: a ``Starshot`` with circle_profile / lines / wobble / angles

    def __len__(self):
        return len(self.status)


class _FrameMeta:
    """What ``Starshot``'s host half asks of its image once the samples are in hand."""

    def __init__(self, shape, dpmm):
        self.shape, self.dpmm = shape, dpmm


def analyze_batch(frames, dpi: float | None = None, sid: float | None = None, radius: float = 0.85,
                  min_peak_height: float = 0.25, max_wobble_diameter: float = 2.0, tolerance: float = 1.0,
                  fwhm: bool = True, recursive: bool = True, invert: bool = False) -> StarshotBatch:
    """``Starshot(frame, dpi=, sid=).analyze(...)`` (pylinac/starshot.py:229-401) for every frame of a uint16 / int16 stack
    [N, H, W] resident in HBM, with the per-frame error cases as status codes.

    Everything that touches pixels runs over the whole stack: the [4, 50, 96] percentiles of the inversion check (exact
    histogram), inversion of the frames that need it, grounding, the axis maxima of the central third and their FW80M
    centres, the 90th percentile, and -- per pass of the retry sweep -- ONE 20-radius ring gather for all frames whose ring
    has the same number of samples (the ring's size follows the start point, so neighbouring frames share it).  The
    profile's tail (roll, 1-D Gaussian, ground, peak search: a few thousand samples) and the reference's own per-dataset
    optimiser (Nelder-Mead over the lines, host scipy: SURVEY section 2 row 7 keeps it out of scope) run per frame on
    ``StarProfile`` / ``Starshot`` themselves, so a frame's numbers are those of the class API."""
    if not 0.2 <= radius <= 0.95:
        raise ValueError("radius must be within (0.2, 0.95)")
    if not 0.05 <= min_peak_height <= 0.95:
        raise ValueError("min_peak_height must be within (0.05, 0.95)")
    x = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames))
    if x.dim() != 3 or x.shape[0] == 0:
        raise ValueError("analyze_batch needs a non-empty [N, H, W] stack")
    if x.dtype not in (torch.uint16, torch.int16):
        raise TypeError("analyze_batch takes uint16 / int16 frames; other types go through Starshot(frame).analyze()")
    if not x.is_cuda:
        x = x.to(_device())
    x = x.contiguous()
    meta = ArrayImage(np.zeros((2, 2), np.uint16), dpi=dpi, sid=sid)
    if meta.dpmm is None:
        raise ValueError("DPI was not a tag in the image nor was it passed in. Please pass a DPI value")
    if meta.sid is None:
        raise ValueError("Source-to-Image distance was not an image tag and was not passed in. Please pass an SID value.")
    dpmm = meta.dpmm
    n, h, w = x.shape
    dev = x.device
    u16 = x.dtype == torch.uint16

    def rows(t, idx):                                     # torch has no indexed copies for uint16: same bits as int16
        sel = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(dev)
        return t.view(torch.int16)[sel].view(torch.uint16) if u16 else t[sel]

    # image.check_inversion_by_histogram([4, 50, 96]) -> ground() -> invert (starshot.py:283-290)
    p = ops.percentile(x, [4, 50, 96]).numpy()
    inverted = np.abs(p[:, 1] - p[:, 0]) > np.abs(p[:, 1] - p[:, 2])
    if inverted.any():
        idx = np.flatnonzero(inverted)
        x = x.clone()
        flipped = ops.invert(rows(x, idx))
        (x.view(torch.int16) if u16 else x)[torch.from_numpy(idx).to(dev)] = flipped.view(torch.int16) if u16 else flipped
    x = ops.ground(x)
    if invert:
        x = ops.invert(x)

    # _get_reasonable_start_point (starshot.py:197-227)
    top_third = int(h / 3)
    bottom_third = int(top_third * 2)
    left_third = int(w / 3)
    right_third = int(left_third * 2)
    xv = x.view(torch.int16) if u16 else x
    central = xv[:, top_third:bottom_third, left_third:right_third].contiguous()
    central = central.view(torch.uint16) if u16 else central
    status = np.zeros(n, dtype=np.int32)
    start = np.full((n, 2), np.nan)
    for col, (axis, shift) in enumerate(((0, left_third), (1, top_third))):
        prof = ops.reduce_axis(central, axis, "max")                               # np.max(central, axis)
        res = ops.find_peaks_batch(prof, cap=1, fwxm_height=0.8, max_number=1)      # FWXMProfile(fwxm_height=80)
        cnt = res.count.cpu().numpy()
        ips = res.props[:, 4:6, 0].cpu().numpy()
        grid = np.arange(prof.shape[1], dtype=float)
        for i in range(n):
            if cnt[i] < 1:
                status[i] = 3
                continue
            left, right = (float(_linear_at(grid, grid, v)) for v in ips[i])        # ProfileBase.x_at_x_idx
            start[i, col] = round(abs(right - left) / 2 + left) + shift             # center_idx (profile.py:322-327)
    local_max = ops.percentile(central, [90]).numpy()[:, 0]

    out = StarshotBatch(status=status, wobble_center=np.full((n, 2), np.nan), wobble_radius=np.full(n, np.nan),
                        wobble_radius_mm=np.full(n, np.nan), wobble_diameter_mm=np.full(n, np.nan),
                        passed=np.zeros(n, dtype=bool), n_lines=np.zeros(n, dtype=np.int32), start_point=start,
                        local_max=local_max, inverted=inverted, radius=np.full(n, np.nan),
                        min_peak_height=np.full(n, np.nan), analyzers=[None] * n)
    shape = (h, w)
    state = {}
    for i in range(n):
        if status[i]:
            continue
        a = object.__new__(Starshot)
        a.image, a.wobble, a.tolerance = _FrameMeta(shape, dpmm), Wobble(), tolerance
        state[i] = [a, Point(x=start[i, 0], y=start[i, 1]), radius, min_peak_height, _retry_sweep(radius, min_peak_height)]

    # _get_reasonable_wobble (starshot.py:306-376): every pass gathers the rings of the frames still looking
    pending = list(state)
    while pending:
        rings, again = {}, []
        for i in pending:
            a, focus, rad, mph, _ = state[i]
            try:
                ring = StarProfile._ring(shape, focus, rad)
            except ValueError:
                again.append(i)
                continue
            rings.setdefault(float(ring.size), []).append((i, ring))
        for size, members in rings.items():
            idx = np.array([i for i, _ in members], dtype=np.int64)
            radii = np.stack([ring._radii for _, ring in members])
            cx = np.array([ring.center.x for _, ring in members], dtype=np.float64)
            cy = np.array([ring.center.y for _, ring in members], dtype=np.float64)
            # profile j on frame idx[j] of the stack: the "combined slices" gather with no neighbours picks frames in place
            vals = ops.circle_profile(x, cx, cy, radii, size, 0, True, 20.0, combine=(idx, n, 0)).cpu().numpy()
            for j, (i, ring) in enumerate(members):
                a, focus, rad, mph, _ = state[i]
                ok = False
                try:
                    ring.values = vals[j]
                    ring.get_peaks(mph * local_max[i], fwhm=fwhm)
                    a.circle_profile = ring
                    ok = a._accept(focus, recursive, max_wobble_diameter)
                except ValueError:
                    ok = False
                except RuntimeError:
                    status[i] = 2
                    continue
                if ok:
                    a.angles = calculate_angles(a.lines)
                    out.analyzers[i] = a
                    out.wobble_center[i] = (a.wobble.center.x, a.wobble.center.y)
                    out.wobble_radius[i], out.wobble_radius_mm[i] = a.wobble.radius, a.wobble.radius_mm
                    out.wobble_diameter_mm[i], out.passed[i] = a.wobble.diameter_mm, a.passed
                    out.n_lines[i], out.radius[i], out.min_peak_height[i] = len(a.lines), rad, mph
                else:
                    again.append(i)
        pending = []
        for i in sorted(again):
            try:
                state[i][2], state[i][3] = next(state[i][4])
                pending.append(i)
            except StopIteration:
                status[i] = 1
    return out
