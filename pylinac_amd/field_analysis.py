"""Flatness / symmetry protocol formulas over ``SingleProfile.field_data`` (pylinac/field_analysis.py:37-231).

Same names and arguments as the reference's module-level calculators; they reduce the in-field values (a few
hundred floats already on the host) to one number each.
"""
from __future__ import annotations

from math import ceil, floor

import numpy as np


def flatness_dose_difference(profile, in_field_ratio: float = 0.8, **kwargs) -> float:
    """Varian flatness (field_analysis.py:37-57)."""
    ser = kwargs.get("slope_exclusion_ratio", 0.2)
    dmax = profile.field_calculation(in_field_ratio=in_field_ratio, calculation="max", slope_exclusion_ratio=ser)
    dmin = profile.field_calculation(in_field_ratio=in_field_ratio, calculation="min", slope_exclusion_ratio=ser)
    return 100 * abs(dmax - dmin) / (dmax + dmin)


def flatness_dose_ratio(profile, in_field_ratio: float = 0.8, **kwargs) -> float:
    """Elekta flatness (field_analysis.py:60-76)."""
    dmax = profile.field_calculation(in_field_ratio=in_field_ratio, calculation="max")
    dmin = profile.field_calculation(in_field_ratio=in_field_ratio, calculation="min")
    return 100 * (dmax / dmin)


def symmetry_point_difference(profile, in_field_ratio: float, **kwargs) -> float:
    """Varian symmetry (field_analysis.py:91-113)."""
    field = profile.field_data(in_field_ratio=in_field_ratio,
                               slope_exclusion_ratio=kwargs.get("slope_exclusion_ratio", 0.2))
    fv = field["field values"]
    cax_value = field["beam center value (@rounded)"]
    sym_vals = [100 * (lt - rt) / cax_value for lt, rt in zip(fv, fv[::-1])]
    return sym_vals[int(np.argmax(np.abs(sym_vals)))]


def symmetry_pdq_iec(profile, in_field_ratio: float, **kwargs) -> float:
    """Elekta PDQ IEC symmetry (field_analysis.py:191-214)."""
    fv = profile.field_data(in_field_ratio=in_field_ratio,
                            slope_exclusion_ratio=kwargs.get("slope_exclusion_ratio", 0.2))["field values"]

    def calc_sym(lt, rt) -> float:
        sym1, sym2 = lt / rt, rt / lt
        sign = np.sign(sym1) if abs(sym1) > abs(sym2) else np.sign(sym2)
        return max(abs(lt / rt), abs(rt / lt)) * sign

    sym_values = [calc_sym(lt, rt) for lt, rt in zip(fv, fv[::-1])]
    return sym_values[int(np.argmax(np.abs(sym_values)))]


def symmetry_area(profile, in_field_ratio: float, **kwargs) -> float:
    """Siemens area symmetry (field_analysis.py:217-231)."""
    fv = profile.field_data(in_field_ratio=in_field_ratio,
                            slope_exclusion_ratio=kwargs.get("slope_exclusion_ratio", 0.2))["field values"]
    n = len(fv)
    area_left = np.sum(fv[: floor(n / 2)])
    area_right = np.sum(fv[ceil(n / 2):])
    return 100 * (area_left - area_right) / (area_left + area_right)


# ---------------------------------------------------------------------------------------------------------------
# Strip profiles and centre search (pylinac/field_analysis.py:488-506, 1068-1117)
# ---------------------------------------------------------------------------------------------------------------
def _strip_edges(length: int, position: float, width: float) -> tuple:
    """The reference's rounding of a strip of relative `width` about relative `position` along an axis of `length`."""
    lo = max(int(round(length * position - length * width / 2)), 0)
    hi = min(int(round(length * position + length * width / 2) + 1), length)
    return lo, hi


def horiz_values(frames, horiz_position: float, horiz_width: float):
    """``FieldAnalysis._get_horiz_values`` (:1094-1117) for a device batch -> (float64 [N, W] profiles
    ``np.mean(array[bottom:top, :], 0)``, bottom_edge, top_edge)."""
    from . import ops

    x = ops._frames(frames)
    bottom, top = _strip_edges(x.shape[1], horiz_position, horiz_width)
    return ops.reduce_axis(x[:, bottom:top, :].contiguous(), 0, "mean"), bottom, top


def vert_values(frames, vert_position: float, vert_width: float):
    """``FieldAnalysis._get_vert_values`` (:1068-1092) -> (float64 [N, H] ``np.mean(array[:, left:right], 1)``, left, right)."""
    from . import ops

    x = ops._frames(frames)
    left, right = _strip_edges(x.shape[2], vert_position, vert_width)
    return ops.reduce_axis(x[:, :, left:right].contiguous(), 1, "mean"), left, right


def determine_center(frame, centering="Beam center") -> tuple:
    """``FieldAnalysis._determine_center`` (:488-506) for one frame -> (vert_ratio, horiz_ratio): the row / column sums
    (device reductions) through ``SingleProfile`` with its defaults."""
    from . import ops
    from .profile import Centering, SingleProfile, _enum

    x = ops._frames(frame)
    if x.shape[0] != 1:
        raise ValueError("determine_center takes one frame")
    vert_sum = ops.reduce_axis(x, 1, "sum")[0].cpu().numpy()
    horiz_sum = ops.reduce_axis(x, 0, "sum")[0].cpu().numpy()
    v_prof, h_prof = SingleProfile(vert_sum), SingleProfile(horiz_sum)
    if _enum(centering, Centering) == Centering.GEOMETRIC_CENTER:
        horiz_ratio = v_prof.geometric_center()["index (exact)"] / x.shape[1]
        vert_ratio = h_prof.geometric_center()["index (exact)"] / x.shape[2]
    else:
        horiz_ratio = v_prof.beam_center()["index (exact)"] / x.shape[1]
        vert_ratio = h_prof.beam_center()["index (exact)"] / x.shape[2]
    return vert_ratio, horiz_ratio
