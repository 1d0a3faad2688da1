"""scikit-image 0.18.3 ``regionprops`` quantities that depend on second moments, formed from EXACT integer raw moments.

The raw moments m00, m10, m01, m20, m02, m11 of a labelled region come from the device as exact 64-bit integers
(``pl_region_moments``).  scikit-image builds the central moments with ``np.dot`` against powers of ``coordinate -
centroid`` (``measure/_moments.py: moments_central``) -- a BLAS summation whose order is implementation-defined -- and
then applies ``inertia_tensor`` / ``orientation`` / ``eccentricity`` (``measure/_regionprops.py:318-322, 394-420``,
``measure/_moments.py: inertia_tensor, inertia_tensor_eigvals``).  Here the central moments are the exact rationals

    mu20 = m20 - m10^2/n,   mu02 = m02 - m01^2/n,   mu11 = m11 - m10*m01/n        (translation invariant)

rounded ONCE to float64, after which scikit-image's own expressions are applied in its operation order.  A region that is
symmetric under the exchange of its axes therefore has ``a - c == 0`` and ``b == 0`` exactly, which is what decides the
+-pi/4 branch of ``orientation`` (called at pylinac/planar_imaging.py:2348, 2498; eccentricity at pylinac/ct.py:2548).
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np


def central_second_moments(m00: int, m10: int, m01: int, m20: int, m02: int, m11: int):
    """exact (mu20/n, mu02/n, mu11/n) as Fractions; rows are axis 0 ("2 0" = row-row)"""
    n = int(m00)
    if n <= 0:
        raise ValueError("empty region")
    m10, m01, m20, m02, m11 = int(m10), int(m01), int(m20), int(m02), int(m11)
    n2 = n * n
    return (Fraction(n * m20 - m10 * m10, n2), Fraction(n * m02 - m01 * m01, n2), Fraction(n * m11 - m10 * m01, n2))


def inertia_tensor(raw) -> np.ndarray:
    """``regionprops.inertia_tensor``: [[mu02, -mu11], [-mu11, mu20]] / mu00"""
    mu20, mu02, mu11 = central_second_moments(*raw)
    a, b, c = float(mu02), -float(mu11), float(mu20)
    return np.array([[a, b], [b, c]], dtype=np.float64)


def orientation(raw) -> float:
    """``regionprops.orientation`` (measure/_regionprops.py:394-402 in 0.18.3)"""
    a, b, b, c = inertia_tensor(raw).flat
    if a - c == 0:
        if b < 0:
            return -math.pi / 4.0
        return math.pi / 4.0
    return 0.5 * math.atan2(-2 * b, c - a)


def inertia_tensor_eigvals(raw):
    ev = np.linalg.eigvalsh(inertia_tensor(raw))
    ev = np.clip(ev, 0, None, out=ev)
    return sorted(ev, reverse=True)


def eccentricity(raw) -> float:
    """``regionprops.eccentricity`` (measure/_regionprops.py:318-322)"""
    l1, l2 = inertia_tensor_eigvals(raw)
    if l1 == 0:
        return 0.0
    return math.sqrt(1 - l2 / l1)


def centroid(raw):
    """``regionprops.centroid`` (row, col): exact quotient of the coordinate sums"""
    n = int(raw[0])
    return (float(Fraction(int(raw[1]), n)), float(Fraction(int(raw[2]), n)))
