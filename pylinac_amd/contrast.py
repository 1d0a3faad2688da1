"""Contrast definitions (pylinac/core/contrast.py:1-137): scalar formulas over ROI statistics that the device computed
(``roi.DiskROI``).  Same names, arguments and error messages as the reference."""
from __future__ import annotations

import enum

import numpy as np


class Contrast(str, enum.Enum):
    """contrast.py:8-15"""

    MICHELSON = "Michelson"
    WEBER = "Weber"
    RATIO = "Ratio"
    RMS = "Root Mean Square"
    DIFFERENCE = "Difference"


def _name(algorithm) -> str:
    return (algorithm.value if isinstance(algorithm, Contrast) else str(algorithm)).lower()


def michelson(array: np.ndarray) -> float:
    """contrast.py:113-120"""
    l_max, l_min = np.nanmax(array), np.nanmin(array)
    return (l_max - l_min) / (l_max + l_min)


def weber(feature: float, background: float) -> float:
    """contrast.py:123-131 (absolute difference, for backwards compatibility)"""
    return abs(feature - background) / background


def ratio(feature: float, reference: float) -> float:
    """contrast.py:134-136"""
    return feature / reference


def difference(feature: float, background: float) -> float:
    """contrast.py:103-110"""
    return abs(feature - background)


def rms(array: np.ndarray) -> float:
    """contrast.py:94-100"""
    if array.min() < 0 or array.max() > 1:
        raise ValueError("RMS calculations require the input array to be normalized. I.e. only values between 0 and 1.")
    return np.sqrt(np.mean((array - array.mean()) ** 2))


def contrast(array: np.ndarray, algorithm) -> float:
    """contrast.py:47-91"""
    a = _name(algorithm)
    array = np.asarray(array)
    if a == "michelson":
        return michelson(array)
    if a == "root mean square":
        return rms(array)
    two = {"weber": ("Weber", "weber", weber), "ratio": ("Ratio", "ratio", ratio),
           "difference": ("Difference", "difference", difference)}
    if a in two:
        label, fn_name, fn = two[a]
        if array.size != 2:
            raise ValueError(f"For {label} algorithm, the array must be exactly 2 elements. Consult the ``{fn_name}`` "
                             "function for parameter details")
        return fn(array[0], array[1])
    raise ValueError(f"Contrast input of {a} did not match any valid options: {[c.value for c in Contrast]}")


def visibility(array: np.ndarray, radius: float, std: float, algorithm) -> float:
    """contrast.py:18-44: the Rose-model visibility ``contrast * sqrt(pi r^2) / std``"""
    return contrast(array, algorithm) * np.sqrt(radius ** 2 * np.pi) / std
