"""2-D gamma index on the device (SURVEY.md section 8 "next" row f4): mirror of ``pylinac.core.gamma.gamma_2d``.

Same arguments, defaults and errors as pylinac/core/gamma.py:229-330; ``reference`` / ``evaluation`` may also be
[N, H, W] batches (one gamma map per pair).  The reference runs a Python loop over every pixel; here one lane
computes one reference pixel (``pl_gamma2d``), bit-identical.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from ._lib import check


def disk_offsets(distance_to_agreement: int):
    """``skimage.draw.disk((0, 0), distance_to_agreement + 1)`` (draw.py ``ellipse`` / ``_ellipse_in_shape``, rotation 0,
    no shape clipping) -> (rr, cc) int arrays in ``np.nonzero`` order."""
    radius = float(distance_to_agreement + 1)
    center = np.array([0.0, 0.0])
    radii = np.array([radius, radius])
    upper_left = np.ceil(center - radii).astype(int)
    lower_right = np.floor(center + radii).astype(int)
    shifted = center - upper_left
    bshape = lower_right - upper_left + 1
    r_lim, c_lim = np.ogrid[0:float(bshape[0]), 0:float(bshape[1])]
    r, c = (r_lim - shifted[0]), (c_lim - shifted[1])
    dist = ((r * 1.0 + c * 0.0) / radii[0]) ** 2 + ((r * 0.0 - c * 1.0) / radii[1]) ** 2
    rr, cc = np.nonzero(dist < 1)
    return rr + upper_left[0], cc + upper_left[1]


def gamma_2d(reference, evaluation, dose_to_agreement: float = 1, distance_to_agreement: int = 1,
             gamma_cap_value: float = 2, global_dose: bool = True, dose_threshold: float = 5,
             fill_value: float = np.nan, device=None) -> torch.Tensor:
    """pylinac/core/gamma.py:229-330 -> float64 tensor with the shape of ``reference``."""
    def dev_of(a):
        return a.device if isinstance(a, torch.Tensor) and a.is_cuda else None

    dev = dev_of(reference) or dev_of(evaluation) or (torch.device(device) if device is not None
                                                      else torch.device("cuda", torch.cuda.current_device()))

    def to_t(a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=dev, dtype=torch.float64).contiguous()

    ref, ev = to_t(reference), to_t(evaluation)
    batched = ref.ndim == 3
    if (ref.ndim, ev.ndim) not in ((2, 2), (3, 3)):
        raise ValueError(f"Reference and evaluation arrays must be 2D. Got reference: {ref.ndim} and evaluation: {ev.ndim}")
    if ref.shape != ev.shape:
        raise ValueError("reference and evaluation must have the same shape")   # the reference would IndexError
    if not batched:
        ref, ev = ref[None], ev[None]
    n, h, w = ref.shape
    dta = int(distance_to_agreement)
    rr, cc = disk_offsets(dta)
    dist2 = (rr / dta) ** 2 + (cc / dta) ** 2            # dist_row**2 + dist_col**2 (gamma.py:296-298)
    d_dr = torch.from_numpy(rr.astype(np.int32)).to(dev)
    d_dc = torch.from_numpy(cc.astype(np.int32)).to(dev)
    d_d2 = torch.from_numpy(np.ascontiguousarray(dist2, dtype=np.float64)).to(dev)
    _, ref_max = ops.minmax(ref)
    work = torch.empty(2 * n * h * w, dtype=torch.float64, device=dev)
    out = torch.empty((n, h, w), dtype=torch.float64, device=dev)
    check(_lib.load().pl_gamma2d(ref.data_ptr(), ev.data_ptr(), n, h, w, dose_to_agreement / 100,
                                 1 if global_dose else 0, ref_max.data_ptr(), d_dr.data_ptr(), d_dc.data_ptr(),
                                 d_d2.data_ptr(), len(rr), dose_threshold / 100, float(gamma_cap_value),
                                 float(fill_value), work.data_ptr(), out.data_ptr(),
                                 torch.cuda.current_stream(dev).cuda_stream), "pl_gamma2d")
    return out if batched else out[0]


def gamma_1d(reference, evaluation, reference_coordinates=None, evaluation_coordinates=None,
             dose_to_agreement: float = 1, distance_to_agreement: int = 1, gamma_cap_value: float = 2,
             global_dose: bool = True, dose_threshold: float = 5, resolution_factor: int = 3,
             fill_value: float = np.nan, device=None):
    """pylinac/core/gamma.py:333-455 -> (gamma, evaluation samples, their x-values) as numpy arrays, like the
    reference (the two sample arrays are the concatenation over the reference points that were evaluated)."""
    reference = np.asarray(reference.cpu() if isinstance(reference, torch.Tensor) else reference)
    evaluation = np.asarray(evaluation.cpu() if isinstance(evaluation, torch.Tensor) else evaluation)
    if reference.ndim != 1 or evaluation.ndim != 1:
        raise ValueError(f"Reference and evaluation arrays must be 1D. Got reference: {reference.ndim} and evaluation: {evaluation.ndim}")
    if reference_coordinates is None:
        reference_coordinates = np.arange(len(reference), dtype=float)
    if len(reference) != len(reference_coordinates):
        raise ValueError(f"Reference and reference_x_values must be the same length. Got reference: {len(reference)} and reference_x_values: {len(reference_coordinates)}")
    if evaluation_coordinates is None:
        evaluation_coordinates = np.arange(len(evaluation), dtype=float)
    if len(evaluation) != len(evaluation_coordinates):
        raise ValueError(f"Evaluation and evaluation_x_values must be the same length. Got evaluation: {len(evaluation)} and evaluation_x_values: {len(evaluation_coordinates)}")
    if min(evaluation_coordinates) - 1 > min(reference_coordinates) or max(evaluation_coordinates) + 1 < max(reference_coordinates):
        raise ValueError("The reference x-values must be within the range of the evaluation x-values")
    if resolution_factor < 1 or not isinstance(resolution_factor, int):
        raise ValueError("Resolution factor must be an integer greater than 0")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ref64 = np.asarray(reference, dtype=np.float64)
    threshold = reference.max() / 100 * dose_threshold
    dose_ta = dose_to_agreement / 100 * reference.max()
    ex = np.asarray(evaluation_coordinates, dtype=np.float64)
    order = np.argsort(ex, kind="mergesort")                    # interp1d(assume_sorted=False) sorts its abscissae
    ex, ev = ex[order], np.asarray(evaluation, dtype=np.float64)[order]
    num = int(distance_to_agreement * resolution_factor * 2 + 1)
    n_ref = len(ref64)

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

    d_ref, d_rx, d_ev, d_ex = up(ref64), up(reference_coordinates), up(ev), up(ex)
    gamma = torch.empty(n_ref, dtype=torch.float64, device=dev)
    vals = torch.empty((n_ref, num), dtype=torch.float64, device=dev)
    xs = torch.empty((n_ref, num), dtype=torch.float64, device=dev)
    computed = torch.empty(n_ref, dtype=torch.int32, device=dev)
    check(_lib.load().pl_gamma1d(d_ref.data_ptr(), d_rx.data_ptr(), n_ref, d_ev.data_ptr(), d_ex.data_ptr(), len(ev),
                                 float(distance_to_agreement), float(distance_to_agreement**2), num, float(threshold),
                                 float(dose_ta), dose_to_agreement / 100, 1 if global_dose else 0,
                                 float(gamma_cap_value), float(fill_value), gamma.data_ptr(), vals.data_ptr(),
                                 xs.data_ptr(), computed.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
          "pl_gamma1d")
    keep = computed.bool().cpu().numpy()
    return gamma.cpu().numpy(), vals.cpu().numpy()[keep].ravel(), xs.cpu().numpy()[keep].ravel()


def gamma_geometric(reference, evaluation, reference_coordinates=None, evaluation_coordinates=None,
                    dose_to_agreement: float = 1, distance_to_agreement: float = 1, gamma_cap_value: float = 2,
                    dose_threshold: float = 5, fill_value: float = np.nan, device=None) -> np.ndarray:
    """pylinac/core/gamma.py:105-227: the geometric (simplex-distance) 1-D gamma of Ju et al. -- for every reference point the
    distance to the piecewise-linear evaluation curve in normalised (position, dose) space; same arguments, validation and
    messages as the reference, one lane per reference point on the device (``pl_gamma_geometric``)."""
    reference, evaluation = np.asarray(reference), np.asarray(evaluation)
    if reference.ndim != 1 or evaluation.ndim != 1:
        raise ValueError(f"Reference and evaluation arrays must be 1D. Got reference: {reference.ndim} and evaluation: {evaluation.ndim}")
    if distance_to_agreement <= 0:
        raise ValueError("Dose to agreement must be greater than 0")          # (the reference swaps the two messages)
    if dose_to_agreement <= 0:
        raise ValueError("Distance to agreement must be greater than 0")

    def monotonic(x):
        d = np.diff(x)
        return bool(np.all(d > 0) or np.all(d < 0))

    if reference_coordinates is None:
        reference_coordinates = np.arange(len(reference), dtype=float)
    reference_coordinates = np.asarray(reference_coordinates)
    if not monotonic(reference_coordinates):
        raise ValueError("Reference x-values must be monotonically increasing or decreasing")
    if len(reference) != len(reference_coordinates):
        raise ValueError(f"Reference and reference_x_values must be the same length. Got reference: {len(reference)} and reference_x_values: {len(reference_coordinates)}")
    if evaluation_coordinates is None:
        evaluation_coordinates = np.arange(len(evaluation), dtype=float)
    evaluation_coordinates = np.asarray(evaluation_coordinates)
    if not monotonic(evaluation_coordinates):
        raise ValueError("Evaluation x-values must be monotonically increasing or decreasing")
    if len(evaluation) != len(evaluation_coordinates):
        raise ValueError(f"Evaluation and evaluation_x_values must be the same length. Got evaluation: {len(evaluation)} and evaluation_x_values: {len(evaluation_coordinates)}")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

    d_ref, d_rx, d_ev, d_ex = up(reference), up(reference_coordinates), up(evaluation), up(evaluation_coordinates)
    out = torch.empty(len(reference), dtype=torch.float64, device=dev)
    decreasing = bool(np.all(np.diff(evaluation_coordinates / distance_to_agreement) < 0))
    check(_lib.load().pl_gamma_geometric(d_ref.data_ptr(), d_rx.data_ptr(), len(reference), d_ev.data_ptr(), d_ex.data_ptr(),
                                         len(evaluation), float(reference.max() * dose_to_agreement), float(distance_to_agreement),
                                         float(dose_threshold) / float(dose_to_agreement), 1 if decreasing else 0,
                                         float(gamma_cap_value), float(fill_value), out.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream), "pl_gamma_geometric")
    return out.cpu().numpy()
