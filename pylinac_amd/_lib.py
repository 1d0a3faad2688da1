"""ctypes binding of libpylinac_hip.so (the C ABI declared in include/pylinac_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the product
raises.  A CPU path would silently void every parity/performance claim.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "libpylinac_hip.so"
_lib = None


class PylinacHipError(RuntimeError):
    pass


class PeakParams(C.Structure):
    """struct pl_peak_params (include/pylinac_hip.h)."""

    _fields_ = [
        ("threshold", C.c_double),
        ("threshold_is_ratio", C.c_int),
        ("distance", C.c_int),
        ("has_prominence", C.c_int),
        ("prominence_min", C.c_double),
        ("width_min", C.c_double),
        ("rel_height", C.c_double),
        ("region_lo", C.c_int),
        ("region_hi", C.c_int),
        ("max_number", C.c_int),
        ("sort_key", C.c_int),
    ]


# dtype / enum values of the header
PL_U16, PL_I16, PL_F32, PL_F64, PL_U8, PL_I32, PL_I64 = 0, 1, 2, 3, 4, 5, 6
PL_SUM, PL_MEAN, PL_MAX, PL_MIN = 0, 1, 2, 3
PL_SORT = {"prominences": 0, "peak_heights": 1, "widths": 2}

ABI_VERSION = 3   # include/pylinac_hip.h: PL_ABI_VERSION (3: pl_pf_measure gained exact_deviation; six round-5 entry points)

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_d = C.c_double

# name -> argtypes ; every symbol declared in include/pylinac_hip.h must be listed here
SIGNATURES = {
    "pl_abi_version": ([], C.c_int),
    "pl_status_string": ([_i], C.c_char_p),
    "pl_last_error": ([], C.c_char_p),
    "pl_device_available": ([], C.c_int),
    "pl_gaussian1d": ([_p, _p, _i, _l, _i, _i, _i, _p, _p, _i, _p], C.c_int),
    "pl_gaussian2d": ([_p, _p, _p, _i, _l, _i, _i, _p, _p, _i, _p], C.c_int),
    "pl_median2d": ([_p, _p, _i, _l, _i, _i, _i, _p], C.c_int),
    "pl_minmax": ([_p, _i, _l, _l, _p, _p, _p], C.c_int),
    "pl_ground": ([_p, _p, _i, _l, _l, _p, _d, _p], C.c_int),
    "pl_normalize": ([_p, _p, _i, _l, _l, _p, _p], C.c_int),
    "pl_invert": ([_p, _p, _i, _l, _l, _p, _p, _p], C.c_int),
    "pl_bit_invert": ([_p, _p, _i, _l, _p], C.c_int),
    "pl_warp_affine": ([_p, _p, _i, _l, _l, _l, _i, _p, _p, _p, _p], C.c_int),
    "pl_scale": ([_p, _p, _i, _l, _l, _d, _p], C.c_int),
    "pl_threshold": ([_p, _p, _i, _l, _l, _p, _i, _i, _p], C.c_int),
    "pl_as_binary": ([_p, _p, _i, _l, _l, _p, _i, _p], C.c_int),
    "pl_hist16": ([_p, _i, _l, _l, _p, _p], C.c_int),
    "pl_hist16_tiles": ([_p, _i, _l, _l, _p, _p, _p], C.c_int),
    "pl_hist16_wl": ([_p, _i, _l, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p], C.c_int),
    "pl_otsu_from_hist": ([_p, _i, _l, _p, _p, _p, _p], C.c_int),
    "pl_otsu16": ([_p, _i, _l, _l, _p, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_median3_otsu16": ([_p, _p, _i, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_order_stats_from_hist": ([_p, _i, _l, _p, _i, _p, _p], C.c_int),
    "pl_reduce_axis": ([_p, _i, _l, _i, _i, _i, _i, _p, _p], C.c_int),
    "pl_threshold_colsum_u16": ([_p, _p, _l, _i, _i, _p, _p, _p], C.c_int),
    "pl_median3_threshold_colsum_u16": ([_p, _p, _l, _i, _i, _p, _p, _p], C.c_int),
    "pl_median3_threshold_colparts_u16": ([_p, _p, _l, _i, _i, _p, _p, _p], C.c_int),
    "pl_colparts_band_rows": ([], C.c_int),
    "pl_colparts_profile_fwxm": ([_p, _l, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_circle_profile": ([_p, _i, _l, _i, _i, _p, _p, _i, _p, _i, _p, _p, _d, _p, _p], C.c_int),
    "pl_circle_profile_combined": ([_p, _i, _l, _i, _i, _p, _l, _l, _i, _p, _p, _i, _p, _i, _p, _p, _d, _p, _p], C.c_int),
    "pl_circle_profile_combined_ex": ([_p, _i, _l, _i, _i, _p, _l, _l, _i, _p, _p, _i, _p, _i, _p, _p, _d, _p, _p, _p], C.c_int),
    "pl_circle_profile_ring": ([_p, _i, _l, _i, _i, _p, _l, _l, _i, _p, _p, _i, _p, _i, _p, _p, _d, _d, _d, _p, _p, _p], C.c_int),
    "pl_phantom_axis_fit": ([_p, _l, _i, _d, _d, _p, _p, _p, _p], C.c_int),
    "pl_sobel": ([_p, _p, _i, _l, _i, _i, _i, _p], C.c_int),
    "pl_label": ([_p, _l, _i, _i, _i, _p, _p, _p, _p], C.c_int),
    "pl_fill_holes": ([_p, _p, _l, _i, _i, _i, _p, _p, _p], C.c_int),
    "pl_binary_centroid": ([_p, _l, _i, _i, _p, _p, _p], C.c_int),
    "pl_scaled_binary": ([_p, _i, _l, _l, _p, _p, _p, _p, _p], C.c_int),
    "pl_field_cax": ([_p, _i, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_field_cax_tiles": ([_p, _i, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_edge_minmax": ([_p, _i, _l, _i, _i, _i, _p, _p, _p], C.c_int),
    "pl_wl_decisions": ([_p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_scharr": ([_p, _p, _i, _l, _i, _i, _p], C.c_int),
    "pl_gaussian2d_mode": ([_p, _p, _p, _i, _l, _i, _i, _p, _i, _i, _p], C.c_int),
    "pl_minmax_masked": ([_p, _p, _l, _l, _p, _p, _p], C.c_int),
    "pl_clip": ([_p, _p, _i, _l, _l, _d, _d, _p], C.c_int),
    "pl_hist_uniform": ([_p, _p, _l, _l, _p, _i, _p, _p], C.c_int),
    "pl_compare": ([_p, _i, _l, _l, _p, _i, _i, _p, _p], C.c_int),
    "pl_clear_border": ([_p, _p, _l, _i, _i, _i, _p, _p, _p], C.c_int),
    "pl_region_stats": ([_p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p], C.c_int),
    "pl_region_moments": ([_p, _l, _i, _i, _i, _p, _p, _p], C.c_int),
    "pl_linspace_edges": ([_p, _p, _i, _l, _p, _p], C.c_int),
    "pl_otsu_from_counts": ([_p, _p, _i, _l, _d, _p, _p, _p], C.c_int),
    "pl_scharr_gaussian": ([_p, _i, _l, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_edge_otsu": ([_p, _i, _p, _i, _l, _i, _i, _p, _i, _p, _p, _p, _p, _d, _p, _p, _p, _p], C.c_int),
    "pl_edge_regions": ([_p, _p, _i, _p, _i, _p, _l, _i, _i, _i, _i, _i, _p, _p, _p, _p, _d, _p, _p, _p], C.c_int),
    "pl_edge_regions_ex": ([_p, _p, _i, _p, _i, _p, _l, _i, _i, _i, _i, _i, _p, _p, _p, _p, _d, _p, _p, _i, _p], C.c_int),
    "pl_edge_otsu_ex": ([_p, _i, _p, _i, _l, _i, _i, _p, _i, _p, _p, _p, _p, _d, _p, _p, _p, _i, _p], C.c_int),
    "pl_peak_valley_regions": ([_p, _l, _i, _l, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_pf_measure": ([_p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _d, _d, _i, _p, _p, _p, _p, _i, _p], C.c_int),
    "pl_scaled_rowmean": ([_p, _l, _i, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p], C.c_int),
    "pl_hill_fit": ([_p, _p, _p, _l, _i, _l, _p, _p, _p, _p, _p], C.c_int),
    "pl_hill_fit_ex": ([_p, _p, _p, _l, _i, _l, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_hill_windows": ([_p, _p, _l, _i, _p, _p, _i, _p, _p, _i, _d, _i, _p, _p, _p, _p, _p], C.c_int),
    "pl_hill_inflection": ([_p, _l, _p, _p], C.c_int),
    "pl_hill_penumbra": ([_p, _p, _l, _d, _d, _p, _p], C.c_int),
    "pl_profile_lookup": ([_p, _p, _l, _i, _p, _i, _p, _p], C.c_int),
    "pl_index_to_original": ([_p, _i, _p, _l, _p, _p], C.c_int),
    "pl_pack_columns": ([_p, _p, _p, _p, _p, _i, _l, _p, _p], C.c_int),
    "pl_edge_plane": ([_p, _i, _l, _i, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p], C.c_int),
    "pl_edge_plane32": ([_p, _i, _l, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_edge_plane32_bracket": ([], C.c_int),
    "pl_edge_plane32_work_bytes": ([_l], C.c_int64),
    "pl_mask_regions_fits": ([_i, _i, _i], C.c_int),
    "pl_mask_regions": ([_p, _i, _p, _l, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p], C.c_int),
    "pl_combine_slices": ([_p, _p, _i, _l, _l, _i, _i, _l, _p], C.c_int),
    "pl_features_level": ([_p, _p, _p, _p, _i, _l, _i, _i, _d, _d, _d, _d, _i, _i, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_features_sweep": ([_p, _l, _i, _i, _d, _d, _d, _d, _i, _p, _i, _p, _p, _p, _p, _p], C.c_int),
    "pl_features_sweep_u16": ([_p, _l, _i, _i, _i, _i, _i, _i, _p, _p, _i, _d, _d, _d, _d, _i, _p, _i, _p, _p, _p, _p, _p],
                              C.c_int),
    "pl_fields_level": ([_p, _p, _p, _i, _l, _i, _i, _d, _d, _d, _d, _i, _i, _i, _p, _p, _p, _p, _p, _p], C.c_int),
    "pl_roi_stats": ([_p, _i, _l, _i, _i, _p, _i, _l, _i, _p, _p, _p], C.c_int),
    "pl_polygon_roi_stats": ([_p, _i, _l, _i, _i, _p, _i, _i, _l, _p, _p, _p], C.c_int),
    "pl_canny_normalise": ([_p, _p, _l, _l, _p, _p], C.c_int),
    "pl_canny_nms": ([_p, _p, _l, _i, _i, _p, _p, _p], C.c_int),
    "pl_canny_mask_prepare": ([_p, _p, _i, _l, _l, _p, _p, _p], C.c_int),
    "pl_canny_nms_masked": ([_p, _p, _l, _i, _i, _p, _i, _p, _p, _p], C.c_int),
    "pl_order_stats_f64": ([_p, _l, _l, _p, _i, _p, _p], C.c_int),
    "pl_hough_line": ([_p, _i, _i, _p, _p, _i, _p, _p], C.c_int),
    "pl_max_filter1d": ([_p, _p, _i, _l, _i, _i, _i, _i, _p], C.c_int),
    "pl_peak_candidates": ([_p, _p, _i, _l, _d, _p, _p], C.c_int),
    "pl_canny_hysteresis": ([_p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _i, _p], C.c_int),
    "pl_xim_work_bytes": ([_i, _i], C.c_int64),
    "pl_xim_decode": ([_p, _l, _p, _l, _i, _i, _i, _p, _p, _p], C.c_int),
    "pl_bakai_mask": ([_p, _p, _l, _l, _p, _p, _p], C.c_int),
    "pl_bakai_gamma": ([_p, _p, _p, _p, C.c_float, C.c_float, _l, _p, _p], C.c_int),
    "pl_gamma1d": ([_p, _p, _i, _p, _p, _i, _d, _d, _i, _d, _d, _d, _i, _d, _d, _p, _p, _p, _p, _p], C.c_int),
    "pl_gamma_geometric": ([_p, _p, _i, _p, _p, _i, _d, _d, _d, _i, _d, _d, _p, _p], C.c_int),
    "pl_gamma2d": ([_p, _p, _l, _i, _i, _d, _i, _p, _p, _p, _p, _i, _d, _d, _d, _p, _p, _p], C.c_int),
    "pl_cast_wrap": ([_p, _p, _i, _l, _p], C.c_int),
    "pl_zoom1d_cubic": ([_p, _l, _i, _i, _i, _p, _p, _p], C.c_int),
    "pl_gradient1d": ([_p, _l, _i, _p, _p], C.c_int),
    "pl_interp1d": ([_p, _l, _p, _l, _i, _p, _i, _i, _p, _p, _p], C.c_int),
    "pl_nps2d": ([_p, _l, _i, _l, _i, _d, _p, _p, _p], C.c_int),
    "pl_nps2d_work_doubles": ([_l, _i], C.c_int64),
    "pl_radial_average": ([_p, _i, _i, _i, _p, _p], C.c_int),
    "pl_esf_mtf": ([_p, _p, _p, _i, _i, _i, _p, _p, _p, _p], C.c_int),
    "pl_to_u16_exact": ([_p, _i, _l, _l, _p, _d, _p, _p, _p], C.c_int),
    "pl_dicom_decode": ([_p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _d, _d, _p, _p], C.c_int),
    "pl_colsum_to_mean": ([_p, _l, _i, _i, _p, _p], C.c_int),
    "pl_find_peaks_var": (
        [_p, _l, _i, _p, _l, C.POINTER(PeakParams), _i, _p, _p, _p, _p, _p, _p, _p],
        C.c_int,
    ),
    "pl_find_peaks_regions": (
        [_p, _l, _i, _p, _l, C.POINTER(PeakParams), _p, _i, _p, _p, _p, _p, _p, _p, _p],
        C.c_int,
    ),
    "pl_scaled_colmean": ([_p, _l, _i, _i, _p, _p, _p, _p], C.c_int),
    "pl_pf_pickets": ([_p, _p, _i, _p, _i, _l, _p, _p, _p, _p], C.c_int),
    "pl_pf_windows": ([_p, _l, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _d, _d, _i, _p, _p, _p, _p, _p],
                      C.c_int),
    "pl_pf_windows_rows": ([_p, _l, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _d, _d, _i, _p, _p, _p, _p, _p],
                      C.c_int),
    "pl_pf_positions": ([_p, _p, _p, _l, _p, _p], C.c_int),
    "pl_fwxm_record": ([_p, _p, _p, _i, _l, _p, _p], C.c_int),
    "pl_find_peaks": (
        [_p, _l, _i, _l, C.POINTER(PeakParams), _i, _p, _p, _p, _p, _p, _p, _p],
        C.c_int,
    ),
}


def lib_path() -> Path:
    return Path(os.environ.get("PYLINAC_HIP_LIB", _LIB_PATH))


def load():
    """Load the shared library (once).  Raises PylinacHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise PylinacHipError(
            f"{path} not found: build it with `python -m pylinac_amd._build` "
            "(or __graft_entry__.build()).  There is no CPU fallback by design."
        )
    # PyTorch bundles its own libamdhip64.so.  Import torch FIRST so that the HIP runtime it loads
    # is the one this library binds to (same soname): device pointers and stream handles passed
    # across the C ABI are only meaningful inside a single runtime instance.
    import torch  # noqa: F401

    lib = C.CDLL(str(path))
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.pl_abi_version() != ABI_VERSION:
        raise PylinacHipError(f"ABI version mismatch: library reports {lib.pl_abi_version()}, this package binds {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        lib = load()
        msg = lib.pl_last_error().decode() or lib.pl_status_string(rc).decode()
        raise PylinacHipError(f"{what or 'libpylinac_hip'} failed (status {rc}): {msg}")
