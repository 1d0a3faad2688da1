"""Noise power spectrum: device mirror of ``pylinac.core.nps``.

Same function names, arguments and error behaviour as pylinac/core/nps.py:12-118; arrays may be numpy or
torch (device) and the heavy steps -- the per-ROI 2-D DFT power and the radial binning -- run through
``pl_nps2d`` / ``pl_radial_average`` (csrc/spectral.hip).  Results are float64 torch tensors on the device.
"""
from __future__ import annotations

import math
from collections.abc import Iterable

import numpy as np
import torch

from . import _lib
from ._lib import check


def _device(dev=None) -> torch.device:
    return torch.device(dev) if dev is not None else torch.device("cuda", torch.cuda.current_device())


def _as_f64(a, dev) -> torch.Tensor:
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=dev, dtype=torch.float64).contiguous()


def radial_average(arr, device=None) -> torch.Tensor:
    """pylinac/core/nps.py:12-32: radial mean about ``floor(shape / 2)``, bins ``int(r)``."""
    dev = arr.device if isinstance(arr, torch.Tensor) and arr.is_cuda else _device(device)
    a = _as_f64(arr, dev)
    if a.ndim != 2:
        raise ValueError("radial_average needs a 2-D array")
    h, w = a.shape
    cy, cx = h // 2, w // 2
    far = max(math.isqrt(dy * dy + dx * dx) for dy in (cy, h - 1 - cy) for dx in (cx, w - 1 - cx))
    nbins = far + 1
    out = torch.empty(nbins, dtype=torch.float64, device=dev)
    check(_lib.load().pl_radial_average(a.data_ptr(), h, w, nbins, out.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream), "pl_radial_average")
    return out


def noise_power_spectrum_2d(pixel_size: float, rois: Iterable, device=None) -> torch.Tensor:
    """pylinac/core/nps.py:35-79.  ``rois``: 2-D arrays (any shapes; the top-left ``length`` square of each is
    used, ``length`` = the smallest dimension over all ROIs) or one [R, H, W] tensor."""
    if isinstance(rois, torch.Tensor) and rois.ndim == 3:
        dev = rois.device if rois.is_cuda else _device(device)
        stack = _as_f64(rois, dev)
        length = int(min(stack.shape[1:]))
    else:
        rois = list(rois)
        if not rois:
            raise ValueError("min() arg is an empty sequence")   # what the reference raises
        dev = next((r.device for r in rois if isinstance(r, torch.Tensor) and r.is_cuda), None) or _device(device)
        length = min(min(r.shape) for r in rois)
        stack = torch.stack([_as_f64(r, dev)[:length, :length] for r in rois]).contiguous()
    n = stack.shape[0]
    lib = _lib.load()
    work = torch.empty(int(lib.pl_nps2d_work_doubles(n, length)), dtype=torch.float64, device=dev)
    out = torch.empty((length, length), dtype=torch.float64, device=dev)
    check(lib.pl_nps2d(stack.data_ptr(), n, length, stack.stride(0), stack.stride(1), float(pixel_size),
                       work.data_ptr(), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "pl_nps2d")
    return out


def noise_power_spectrum_1d(spectrum_2d) -> torch.Tensor:
    """pylinac/core/nps.py:82-96 (``validators.double_dimension`` -> ValueError)."""
    if spectrum_2d.ndim != 2:
        raise ValueError(f"Array was not 2D. Got shape: {tuple(spectrum_2d.shape)}")
    return radial_average(spectrum_2d)


def _np1d(nps1d) -> np.ndarray:
    a = nps1d.detach().cpu().numpy() if isinstance(nps1d, torch.Tensor) else np.asarray(nps1d)
    if a.ndim != 1:
        raise ValueError(f"Array was not 1D. Got shape: {a.shape}")
    return a


def average_power(nps1d) -> float:
    """pylinac/core/nps.py:99-115: power-weighted mean of ``linspace(0, 1, len)`` (a few hundred floats)."""
    a = _np1d(nps1d)
    return float(np.average(np.linspace(0, 1, len(a)), weights=a))


def max_frequency(nps1d) -> float:
    """pylinac/core/nps.py:118-121."""
    a = _np1d(nps1d)
    return float(np.argmax(a) / len(a))
