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
from itertools import product

import numpy as np
from scipy import optimize

from . import ops
from .array_utils import _Staged
from .image import ArrayImage
from .profile import CollapsedCircleProfile, FWXMProfile, Point


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

    def get_peaks(self, min_peak_height, min_peak_distance=0.02, fwhm: bool = True):
        roll_amount = np.where(self.values == self.values.min())[0][0]
        self.roll(roll_amount)
        self.filter(size=0.003, kind="gaussian")
        self.ground()
        if fwhm:
            self.find_fwxm_peaks(threshold=min_peak_height, min_distance=min_peak_distance)
        else:
            self.find_peaks(min_peak_height, min_peak_distance)


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
        peak_heights = np.append(min_peak_height, np.linspace(0.05, 0.95, 10))
        radii = np.append(radius, np.linspace(0.95, 0.1, 10))
        radius_and_peak_gen = product(radii, peak_heights)
        while True:
            try:
                min_height = min_peak_height * local_max
                self.circle_profile = StarProfile(self.image, focus_point, radius, min_height, fwhm)
                n = len(self.circle_profile.peaks)
                if n < 6 or n % 2 != 0:
                    if not recursive:
                        raise RuntimeError("The algorithm was unable to properly detect the radiation lines. Try setting "
                                           "recursive to True or lower the minimum peak height")
                    raise ValueError
                self.lines = LineManager(self.circle_profile.peaks, focus_point=focus_point, dpmm=self.image.dpmm)
                self._find_wobble_minimize()
                focus_near_center = _distance(self.wobble.center, focus_point) < 10 * self.image.dpmm
                if (self.wobble.diameter_mm < max_wobble_diameter and focus_near_center) or not recursive:
                    return
                raise ValueError
            except ValueError:
                try:
                    radius, min_peak_height = next(radius_and_peak_gen)
                except StopIteration:
                    raise RuntimeError("The algorithm was unable to determine a reasonable wobble. Try setting "
                                       "recursive to False and manually adjusting algorithm parameters")

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
