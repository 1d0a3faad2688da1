"""Peak-valley (Michelson) relative MTF.

Mirror of ``pylinac.core.mtf.MTF`` (pylinac/core/mtf.py:32-101) and of the CTP528 driver
``CTP528CP504.mtf`` (pylinac/ct.py:1511-1544).  The arithmetic here is a handful of scalar
operations on <= 8 line-pair regions per slice (SURVEY: "negligible"), so it is host-side numpy like
the reference; the heavy part it consumes -- the collapsed circle profile and its peak/valley
searches -- runs on the GPU (``profile.CollapsedCircleProfile``, ``find_peaks``).
"""
from __future__ import annotations

import warnings
from collections.abc import Sequence

import numpy as np


def michelson(array: np.ndarray) -> float:
    """pylinac/core/contrast.py:108-116: (max - min) / (max + min)."""
    l_max, l_min = np.nanmax(array), np.nanmin(array)
    return (l_max - l_min) / (l_max + l_min)


def _interp_linear_extrapolate(xp, fp, x: float) -> float:
    """``scipy.interpolate.interp1d(xp, fp, fill_value="extrapolate")(x)`` (linear): the segment is
    found with searchsorted on the SORTED abscissae, slope form ``slope * (x - x_lo) + y_lo``."""
    xp = np.asarray(xp, dtype=float)
    fp = np.asarray(fp, dtype=float)
    order = np.argsort(xp, kind="mergesort")   # interp1d sorts its x (assume_sorted=False)
    xp, fp = xp[order], fp[order]
    hi = int(np.clip(np.searchsorted(xp, x), 1, len(xp) - 1))
    lo = hi - 1
    slope = (fp[hi] - fp[lo]) / (xp[hi] - xp[lo])
    return float(slope * (x - xp[lo]) + fp[lo])


class MTF:
    """pylinac/core/mtf.py:32-101."""

    def __init__(self, lp_spacings: Sequence[float], lp_maximums: Sequence[float], lp_minimums: Sequence[float]):
        self.spacings = lp_spacings
        self.maximums = lp_maximums
        self.minimums = lp_minimums
        if len(lp_spacings) != len(lp_maximums) != len(lp_minimums):
            raise ValueError("The number of MTF spacings, maximums, and minimums must be equal.")
        if len(lp_spacings) < 2 or len(lp_maximums) < 2 or len(lp_minimums) < 2:
            raise ValueError("The number of MTF spacings, maximums, and minimums must be greater than 1.")
        self.mtfs = {}
        self.norm_mtfs = {}
        for spacing, mx, mn in zip(lp_spacings, lp_maximums, lp_minimums):
            self.mtfs[spacing] = michelson(np.array((mx, mn)))
        self.mtfs = {k: v for k, v in sorted(self.mtfs.items(), key=lambda x: x[0])}
        for key, value in self.mtfs.items():
            self.norm_mtfs[key] = value / self.mtfs[lp_spacings[0]]
        if np.max(np.diff(list(self.norm_mtfs.values()))) > 0:
            warnings.warn("The MTF does not drop monotonically; be sure the ROIs are correctly aligned.")

    def relative_resolution(self, x: float = 50) -> float:
        """Line-pair value at the given rMTF percentage (mtf.py:82-101)."""
        if not 0 <= x <= 100:
            raise ValueError("x must be within (0, 100)")
        mtf = _interp_linear_extrapolate(list(self.norm_mtfs.values()), list(self.norm_mtfs.keys()), x / 100)
        if mtf > max(self.spacings):
            warnings.warn(f"MTF resolution wasn't calculated for {x}% that was asked for. The value returned is an "
                          "extrapolation. Use a higher % MTF to get a non-interpolated value.")
        return float(mtf)


def peak_valley_mtf(circle_profile, roi_settings: dict) -> MTF:
    """``CTP528CP504.mtf`` (pylinac/ct.py:1511-1544): per line-pair region the mean of the
    ``num peaks`` highest peaks and of the valleys between them, Michelson contrast, normalised to
    region 1.  ``circle_profile`` is a (Collapsed)CircleProfile / MultiProfile of this package."""
    maxs, mins = [], []
    for value in roi_settings.values():
        max_indices, max_values = circle_profile.find_peaks(
            min_distance=value["peak spacing"], max_number=value["num peaks"],
            search_region=(value["start"], value["end"]))
        if len(max_values) != value["num peaks"]:
            break
        maxs.append(max_values.mean())
        _, min_values = circle_profile.find_valleys(
            min_distance=value["peak spacing"], max_number=value["num valleys"],
            search_region=(min(max_indices), max(max_indices)))
        mins.append(min_values.mean())
    if not maxs:
        raise ValueError("Did not find any spatial resolution pairs to analyze.")
    spacings = [roi["lp/mm"] for roi in roi_settings.values()]
    return MTF(lp_spacings=spacings, lp_maximums=maxs, lp_minimums=mins)


# ---------------------------------------------------------------------------- ESF-FFT MTF (a18)
def hann(m: int) -> np.ndarray:
    """``scipy.signal.windows.hann(m)`` (symmetric), restated: general_cosine with a = [0.5, 0.5] over
    ``linspace(-pi, pi, m)`` (scipy/signal/windows/_windows.py, not vendored in the reference tree)."""
    if m <= 1:
        return np.ones(max(m, 0))
    fac = np.linspace(-np.pi, np.pi, m)
    w = np.zeros(m)
    for k, a in enumerate((0.5, 0.5)):
        w += a * np.cos(k * fac)
    return w


def boxcar(m: int) -> np.ndarray:
    return np.ones(m)


class EdgeSpreadFunctionMTF:
    """pylinac/core/mtf.py:308-376: relative MTF from edge spread functions, averaged over the ESFs.

    Same constructor arguments (``esf, sample_spacing, padding_mode, num_samples, windowing, **kwargs``) and
    attributes (``freq``, ``mtf``, ``relative_resolution``).  ``windowing`` is any callable ``f(len, **kwargs)``
    (default: the Hann window); gradient, windowing, the zero-padded DFT magnitude, normalisation and the mean
    over ESFs run on the GPU (``pl_esf_mtf``).
    """

    def __init__(self, esf, sample_spacing: float | None = None, padding_mode: str = "auto",
                 num_samples: int = 1024, windowing=hann, device=None, **kwargs):
        import torch

        from . import _lib
        from ._lib import check

        self.sample_spacing = sample_spacing
        windowing = windowing or boxcar
        esf = [np.asarray(e.detach().cpu().numpy() if hasattr(e, "detach") else e, dtype=np.float64) for e in esf]
        len_esf = np.unique([len(e) for e in esf])
        if padding_mode == "none":
            if len(len_esf) > 1:
                raise ValueError("If padding_mode='none', all ESF samples must have the same size")
            num_samples = int(len_esf[0])
        elif padding_mode == "fixed":
            if num_samples < max(len_esf):
                raise ValueError("num_samples must be larger than the largest array")
        elif padding_mode == "auto":
            next_power_of_two = max(2 ** np.ceil(np.log2(len_esf)))
            num_samples = int(max(next_power_of_two, num_samples))
        if min(len_esf) < 2:
            raise ValueError("Shape of array too small to calculate a numerical gradient, "
                             "at least (edge_order + 1) elements are required.")   # np.gradient's message
        pixel_spacing = 1 if sample_spacing is None else sample_spacing
        # np.fft.fftfreq(n, d)[: n // 2] = k / (n * d), computed as numpy does (integer k times 1/(n d))
        self.freq = np.arange(0, num_samples // 2, dtype=int) * (1.0 / (num_samples * pixel_spacing))

        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        n_esf, lmax, half = len(esf), int(max(len_esf)), num_samples // 2
        host = np.zeros((2, n_esf, lmax))
        for i, e in enumerate(esf):
            host[0, i, : len(e)] = e
            host[1, i, : len(e)] = np.asarray(windowing(len(e), **kwargs), dtype=np.float64)
        both = torch.from_numpy(host).to(dev)
        lens = torch.tensor([len(e) for e in esf], dtype=torch.int32, device=dev)
        work = torch.empty((n_esf, half), dtype=torch.float64, device=dev)
        each = torch.empty((n_esf, half), dtype=torch.float64, device=dev)
        mean = torch.empty(half, dtype=torch.float64, device=dev)
        check(_lib.load().pl_esf_mtf(both[0].data_ptr(), lens.data_ptr(), both[1].data_ptr(), n_esf, lmax,
                                     num_samples, work.data_ptr(), each.data_ptr(), mean.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream), "pl_esf_mtf")
        self._mtf = list(each.cpu().numpy())
        self._esf = esf
        self.mtf = mean.cpu().numpy()

    def relative_resolution(self, x: float = 50) -> float:
        """pylinac/core/mtf.py:378-388 (``argue.bounds(x=(0, 100))`` -> ValueError)."""
        if not 0 <= x <= 100:
            raise ValueError("x must be within (0, 100)")
        return float(np.interp(-x / 100, -self.mtf, self.freq))
