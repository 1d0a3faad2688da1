"""Host-side value types the profile / image classes hand to the analyzers: the members of
``pylinac.core.geometry.Point`` / ``Vector`` / ``Circle`` (pylinac/core/geometry.py:70-205, 215-247, 399-473) that analyzer
code touches on objects it gets FROM a profile or an image -- ``copy.copy(profile.center)``, ``center.as_array()``,
``peak.x``, ``point.distance_to(other)``, ``Line(peak_a, peak_b)`` (which reads ``as_array()`` of its end points),
``circle.diameter``.  No arithmetic of the hot path lives here; plotting is out of scope.

A :class:`Point` iterates as ``(x, y, z, idx, value)`` so that the reference's own copy constructor ``Point(thing)`` (its
iterable branch) takes every field of one, and it accepts the reference's points by attribute: analyzers may mix the two.
"""
from __future__ import annotations

import math

import numpy as np

_FIELDS = ("x", "y", "z", "idx", "value")


def _coords(thing) -> tuple:
    """(x, y, z) of a point-like: anything with .x / .y (z optional) or an iterable of up to three numbers"""
    if hasattr(thing, "x") and hasattr(thing, "y"):
        return thing.x, thing.y, getattr(thing, "z", 0) or 0
    c = tuple(thing) + (0, 0, 0)
    return c[0], c[1], c[2]


class Point:
    """geometry.py:70-205.  ``Point(3, 4)``, ``Point((3, 4))``, ``Point(other_point)``; ``idx`` / ``value`` describe a sample of
    a profile (a peak's index and height)."""

    __slots__ = _FIELDS

    def __init__(self, x=0, y=0, z=0, idx=None, value=None, as_int: bool = False):
        if hasattr(x, "x") and hasattr(x, "y"):                       # a point of either package: every field it has
            self.x, self.y, self.z = _coords(x)
            self.idx, self.value = getattr(x, "idx", None), getattr(x, "value", None)
        elif np.iterable(x):                                          # (x, y[, z[, idx[, value]]]), missing -> 0
            items = list(x) + [0] * len(_FIELDS)
            for name, item in zip(_FIELDS, items):
                setattr(self, name, item)
        else:
            self.x, self.y, self.z, self.idx, self.value = x, y, z, idx, value
        if as_int:
            self.x, self.y, self.z = int(round(self.x)), int(round(self.y)), int(round(self.z))

    def __iter__(self):
        # all five fields, in the order the reference's copy constructor zips them (geometry.py:108-110)
        return iter((self.x, self.y, self.z, self.idx, self.value))

    def __copy__(self):
        return Point(self)

    def __deepcopy__(self, memo):
        return Point(self)

    def distance_to(self, thing) -> float:
        """Euclidean distance to a point-like; to a circle: distance from its perimeter (geometry.py:122-138)"""
        if hasattr(thing, "center") and hasattr(thing, "radius"):
            return abs(np.sqrt((self.x - thing.center.x) ** 2 + (self.y - thing.center.y) ** 2) - thing.radius)
        px, py, pz = _coords(thing)
        return math.sqrt((self.x - px) ** 2 + (self.y - py) ** 2 + (self.z - pz) ** 2)

    def as_array(self, coords=("x", "y", "z")) -> np.ndarray:
        return np.array([getattr(self, c) for c in coords])

    def as_vector(self) -> "Vector":
        return Vector(self.x, self.y, self.z)

    def dict(self) -> dict:
        return {k: float(getattr(self, k)) for k in _FIELDS if getattr(self, k) is not None}

    def __repr__(self) -> str:
        return f"Point(x={self.x:3.2f}, y={self.y:3.2f}, z={self.z:3.2f})"

    def __eq__(self, other) -> bool:
        return all(getattr(self, k) == getattr(other, k, None) for k in _FIELDS)

    __hash__ = None

    def _combine(self, other, op) -> "Vector":
        out = Vector()
        for k in _FIELDS:                                             # idx / value too; None where they do not combine
            try:
                setattr(out, k, op(getattr(self, k), getattr(other, k)))
            except (TypeError, AttributeError):
                setattr(out, k, None)
        return out

    def __add__(self, other) -> "Vector":
        return self._combine(other, lambda a, b: a + b)

    def __sub__(self, other) -> "Vector":
        return self._combine(other, lambda a, b: a - b)

    def __mul__(self, factor):                                        # in place, like the reference
        for k in _FIELDS:
            if getattr(self, k) is not None:
                setattr(self, k, getattr(self, k) * factor)
        return self

    def __truediv__(self, divisor):
        for k in _FIELDS:
            if getattr(self, k) is not None:
                setattr(self, k, getattr(self, k) / divisor)
        return self


class Vector:
    """geometry.py:408-473"""

    def __init__(self, x=0, y=0, z=0):
        self.x, self.y, self.z = x, y, z

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __repr__(self) -> str:
        return f"Vector(x={self.x:.2f}, y={self.y:.2f}, z={self.z:.2f})"

    def as_scalar(self) -> float:
        return math.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2)

    def as_point(self) -> Point:
        return Point(self.x, self.y, self.z)

    def dict(self) -> dict:
        return {"x": self.x, "y": self.y, "z": self.z}

    def distance_to(self, thing) -> float:
        return Point(self.x, self.y, self.z).distance_to(thing)

    def __sub__(self, other) -> "Vector":
        return Vector(self.x - other.x, self.y - other.y, self.z - other.z)

    def __add__(self, other) -> "Vector":
        return Vector(self.x + other.x, self.y + other.y, self.z + other.z)

    def __neg__(self) -> "Vector":
        return Vector(-self.x, -self.y, -self.z)

    def __truediv__(self, divisor) -> "Vector":
        self.x, self.y, self.z = self.x / divisor, self.y / divisor, self.z / divisor
        return self


class Circle:
    """geometry.py:215-247, 399-405: the base of ``CircleProfile`` (``center``, ``radius``, ``area``, ``diameter``, ``as_dict``)"""

    def __init__(self, center_point=(0, 0), radius: float = 0):
        if center_point is None:
            center_point = Point()
        elif not (hasattr(center_point, "x") or np.iterable(center_point)):
            raise TypeError("Circle center must be of type Point or iterable")
        self.center = Point(center_point)
        self.radius = radius

    @property
    def area(self) -> float:
        return math.pi * self.radius ** 2

    @property
    def diameter(self) -> float:
        return self.radius * 2

    def as_dict(self) -> dict:
        return {"center_x": self.center.x, "center_y": self.center.y, "diameter": self.diameter}
