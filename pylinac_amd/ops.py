"""Batched device operators: thin, validating wrappers over the C ABI.

Everything here takes and returns ``torch`` tensors resident on the GPU, laid out ``[N, H, W]``
(frames) or ``[N, L]`` (profiles).  PyTorch is used for device memory and streams only; all
arithmetic is in libpylinac_hip.so.  The numpy-facing mirror of the reference API
(``pylinac_amd.array_utils`` / ``.image`` / ``.profile``) is built on these functions.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import PL_F32, PL_F64, PL_I16, PL_I32, PL_I64, PL_U8, PL_U16, PeakParams, check

_DTYPES = {torch.uint16: PL_U16, torch.int16: PL_I16, torch.float32: PL_F32, torch.float64: PL_F64,
           torch.uint8: PL_U8, torch.int32: PL_I32, torch.int64: PL_I64}
_REDUCE = {"sum": _lib.PL_SUM, "mean": _lib.PL_MEAN, "max": _lib.PL_MAX, "min": _lib.PL_MIN}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class HostCopy:
    """A device-to-host copy that does not stop the host: the tensor is copied into pinned memory on the current stream and
    an event marks its arrival; ``numpy()`` waits for that event only.  Lets a caller queue the next batch's kernels while
    an earlier batch's scalars are on their way (ct.ctp528_batch)."""

    def __init__(self, t: torch.Tensor):
        if t.device.type == "cuda":
            self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._host.copy_(t, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream())
        else:                                   # (CPU tensors only reach this under the test suite's emulated device)
            self._host = t.clone()
            self._event = None

    def numpy(self) -> np.ndarray:
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return self._host.numpy()


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(
            f"unsupported dtype {t.dtype}; supported: uint8, uint16, int16, int32, int64, float32, float64"
        ) from None


def _frames(t: torch.Tensor) -> torch.Tensor:
    """Validate a device batch [N,H,W] (a single [H,W] frame is viewed as N=1)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor on a HIP device")
    if not t.is_cuda:
        raise ValueError("tensor must live on the GPU (no CPU fallback exists)")
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3:
        raise ValueError(f"expected [N,H,W] or [H,W]; got shape {tuple(t.shape)}")
    if t.numel() == 0:
        raise ValueError("Array must not be empty")
    return t.contiguous()


def _per_frame(v, n: int, device) -> tuple[torch.Tensor, int]:
    """Scalar or per-frame values -> float64 device tensor + stride (0 broadcast / 1 per frame)."""
    if isinstance(v, torch.Tensor):
        v = v.to(device=device, dtype=torch.float64).reshape(-1).contiguous()
        if v.numel() == 1:
            return v, 0
        if v.numel() != n:
            raise ValueError(f"expected 1 or {n} per-frame values; got {v.numel()}")
        return v, 1
    return torch.tensor([float(v)], dtype=torch.float64, device=device), 0


# ------------------------------------------------------------------------------------ filtering
def gaussian_weights(sigma: float, truncate: float = 4.0) -> tuple[np.ndarray, int]:
    """scipy's ``_gaussian_kernel1d`` (order 0) as used by ``ndimage.gaussian_filter``:
    radius ``int(truncate*sigma + 0.5)``; host float64, formula and evaluation order identical
    (scipy/ndimage/_filters.py)."""
    sd = float(sigma)
    lw = int(truncate * sd + 0.5)
    x = np.arange(-lw, lw + 1)
    phi = np.exp(-0.5 / (sd * sd) * x**2)
    phi = phi / phi.sum()
    return phi[::-1].copy(), lw


_weights_cache: dict = {}


def _device_weights(sigma: float, device) -> tuple[torch.Tensor, np.ndarray, int]:
    """(taps on the device, the same taps in host memory, radius): the C ABI takes both copies (pylinac_hip.h)"""
    key = (float(sigma), str(device))
    hit = _weights_cache.get(key)
    if hit is None:
        w, lw = gaussian_weights(sigma)
        w = np.ascontiguousarray(w, dtype=np.float64)
        hit = (torch.from_numpy(w).to(device), w, lw)
        _weights_cache[key] = hit
    return hit


def gaussian_filter(frames: torch.Tensor, sigma: float, out=None, tmp=None) -> torch.Tensor:
    """``ndimage.gaussian_filter(frame, sigma)`` per frame (pylinac/core/array_utils.py:133)."""
    x = _frames(frames)
    n, h, w = x.shape
    wts, hw, lw = _device_weights(sigma, x.device)
    out = torch.empty_like(x) if out is None else out
    tmp = torch.empty_like(x) if tmp is None else tmp
    check(
        _lib.load().pl_gaussian2d(x.data_ptr(), out.data_ptr(), tmp.data_ptr(), _dt(x), n, h, w,
                                  wts.data_ptr(), hw.ctypes.data, lw, _stream()),
        "pl_gaussian2d",
    )
    return out


def gaussian_filter1d(x: torch.Tensor, sigma: float, axis: int = -1) -> torch.Tensor:
    """One correlate1d pass.  ``x``: [N,L] profiles (axis=-1) or [N,H,W] frames (axis 0 / 1)."""
    if x.dim() == 2 and axis in (-1, 1):
        f = _frames(x.unsqueeze(1))  # [N,1,L]
        ax = 1
    else:
        f = _frames(x)
        ax = axis
    n, h, w = f.shape
    wts, hw, lw = _device_weights(sigma, f.device)
    out = torch.empty_like(f)
    check(
        _lib.load().pl_gaussian1d(f.data_ptr(), out.data_ptr(), _dt(f), n, h, w, ax, wts.data_ptr(), hw.ctypes.data,
                                  lw, _stream()),
        "pl_gaussian1d",
    )
    return out.reshape(x.shape)


def median_filter(frames: torch.Tensor, size: int, out=None) -> torch.Tensor:
    """``ndimage.median_filter(frame, size=size)`` per frame (pylinac/core/array_utils.py:131).
    [N,L] input is filtered as N 1-D profiles."""
    x = _frames(frames)
    n, h, w = x.shape
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().pl_median2d(x.data_ptr(), out.data_ptr(), _dt(x), n, h, w, int(size), _stream()),
          "pl_median2d")
    return out


def median_filter1d(profiles: torch.Tensor, size: int) -> torch.Tensor:
    x = profiles if profiles.dim() == 2 else profiles.unsqueeze(0)
    return median_filter(x.unsqueeze(1), size).reshape(profiles.shape)


# -------------------------------------------------------------------------- min/max + mutators
def minmax(frames: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    x = _frames(frames)
    n = x.shape[0]
    mn = torch.empty(n, dtype=torch.float64, device=x.device)
    mx = torch.empty(n, dtype=torch.float64, device=x.device)
    check(_lib.load().pl_minmax(x.data_ptr(), _dt(x), n, x[0].numel(), mn.data_ptr(), mx.data_ptr(), _stream()),
          "pl_minmax")
    return mn, mx


def ground(frames: torch.Tensor, value: float = 0.0, mn=None) -> torch.Tensor:
    x = _frames(frames)
    n = x.shape[0]
    if mn is None:
        mn, _ = minmax(x)
    mn = _per_frame(mn, n, x.device)[0].expand(n).contiguous()     # raw pointer below: dense float64 [N]
    out = torch.empty_like(x)
    check(_lib.load().pl_ground(x.data_ptr(), out.data_ptr(), _dt(x), n, x[0].numel(), mn.data_ptr(),
                                float(value), _stream()), "pl_ground")
    return out


def normalize(frames: torch.Tensor, value=None) -> torch.Tensor:
    """``array / val`` -> float64 (float32 frames stay float32, like numpy)."""
    x = _frames(frames)
    n = x.shape[0]
    if value is None:
        _, val = minmax(x)
    else:
        val, stride = _per_frame(value, n, x.device)
        if x.dtype == torch.float32:  # numpy rounds a python scalar to the array dtype (NEP 50)
            val = val.to(torch.float32).to(torch.float64)
        if stride == 0:
            val = val.expand(n).contiguous()
    out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    check(_lib.load().pl_normalize(x.data_ptr(), out.data_ptr(), _dt(x), n, x[0].numel(), val.data_ptr(),
                                   _stream()), "pl_normalize")
    # float32 / float32 is float32 in numpy; rounding the float64 quotient once more is exact
    return out.to(torch.float32) if x.dtype == torch.float32 else out


def invert(frames: torch.Tensor) -> torch.Tensor:
    x = _frames(frames)
    n = x.shape[0]
    mn, mx = minmax(x)
    out = torch.empty_like(x)
    check(_lib.load().pl_invert(x.data_ptr(), out.data_ptr(), _dt(x), n, x[0].numel(), mn.data_ptr(),
                                mx.data_ptr(), _stream()), "pl_invert")
    return out


def to_u16_exact(frames: torch.Tensor, max_range: float | None = None):
    """``frame - frame.min()`` as uint16 where that is exact (``pl_to_u16_exact``): int16 / int32 / float64 frames whose
    values are integers spanning at most ``max_range`` (default: 32767 for int16 -- beyond it the reference's own int16
    ``ground()`` wraps around --, 65535 otherwise).  -> (uint16 frames, flag int32 [N]: 1 = frame does not qualify)."""
    x = _frames(frames)
    if x.dtype not in (torch.int16, torch.int32, torch.float64):
        raise TypeError("to_u16_exact: int16, int32 or float64 frames")
    n = x.shape[0]
    if max_range is None:
        max_range = 32767.0 if x.dtype == torch.int16 else 65535.0
    mn, _ = minmax(x)
    out = torch.empty(x.shape, dtype=torch.uint16, device=x.device)
    flag = torch.empty(n, dtype=torch.int32, device=x.device)
    check(_lib.load().pl_to_u16_exact(x.data_ptr(), _dt(x), n, x[0].numel(), mn.data_ptr(), float(max_range), out.data_ptr(),
                                      flag.data_ptr(), _stream()), "pl_to_u16_exact")
    return out, flag


def scale(frames: torch.Tensor, factor: float) -> torch.Tensor:
    """``array * scalar`` in the array's dtype (the multiply inside ``stretch``)."""
    x = _frames(frames)
    out = torch.empty_like(x)
    check(_lib.load().pl_scale(x.data_ptr(), out.data_ptr(), _dt(x), x.shape[0], x[0].numel(), float(factor),
                               _stream()), "pl_scale")
    return out


def threshold(frames: torch.Tensor, thr, kind: str = "high", out=None) -> torch.Tensor:
    """``np.where(a >= t, a, 0)`` ('high') / ``np.where(a <= t, a, 0)`` (pylinac/core/image.py:797-800)."""
    x = _frames(frames)
    n = x.shape[0]
    t, stride = _per_frame(thr, n, x.device)
    if x.dtype == torch.float32 and not isinstance(thr, (torch.Tensor, np.generic)):
        t = t.to(torch.float32).to(torch.float64)  # NEP 50: python scalar adopts the array dtype
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().pl_threshold(x.data_ptr(), out.data_ptr(), _dt(x), n, x[0].numel(), t.data_ptr(),
                                   stride, 0 if kind == "high" else 1, _stream()), "pl_threshold")
    return out


def as_binary(frames: torch.Tensor, thr) -> torch.Tensor:
    """``a >= t`` as uint8 0/1 (pylinac/core/image.py:802-815; the host mirror widens to int64)."""
    x = _frames(frames)
    n = x.shape[0]
    t, stride = _per_frame(thr, n, x.device)
    if x.dtype == torch.float32 and not isinstance(thr, (torch.Tensor, np.generic)):
        t = t.to(torch.float32).to(torch.float64)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.load().pl_as_binary(x.data_ptr(), out.data_ptr(), _dt(x), n, x[0].numel(), t.data_ptr(),
                                   stride, _stream()), "pl_as_binary")
    return out


# ----------------------------------------------------------------- histogram / Otsu / percentile
def histogram16(frames: torch.Tensor, out=None, tiles: bool = False, edge_window: int | None = None, ranks=None):
    """Exact per-frame histogram of a 16-bit integer batch: uint32 [N, 65536] (stored in an int32
    tensor; bin b = value b for uint16, value b-32768 for int16).  ``tiles=True`` -> (histogram, tile maxima): the same pass
    also leaves the largest key of every 512-pixel tile of every frame (uint16 [N, ceil(pixels / 512)], stored in an int16
    tensor; 0xffff = "look inside"), which :func:`field_cax` uses to skip the tiles that cannot hold foreground;
    ``edge_window=k`` (with ``tiles``) -> (histogram, tile maxima, edge min, edge max): :func:`edge_minmax` in the same launch;
    with ``ranks`` too -> (None, tile maxima, edge min, edge max, order statistics int32 [N, len(ranks)]): :func:`order_stats`
    selected inside the histogram launch -- the table itself is scratch then."""
    x = _frames(frames)
    if x.dtype not in (torch.uint16, torch.int16):
        raise TypeError("histogram16 needs uint16 or int16 frames")
    n = x.shape[0]
    out = torch.empty((n, 65536), dtype=torch.int32, device=x.device) if out is None else out
    if not tiles:
        check(_lib.load().pl_hist16(x.data_ptr(), _dt(x), n, x[0].numel(), out.data_ptr(), _stream()), "pl_hist16")
        return out
    tmax = torch.empty((n, (x[0].numel() + 511) // 512), dtype=torch.int16, device=x.device)
    if edge_window is not None:       # + edge_minmax(frames, edge_window) in the same launch -> (hist, tiles, edge min, edge max)
        emin = torch.empty(n, dtype=torch.int32, device=x.device)
        emax = torch.empty_like(emin)
        if ranks is not None:         # + order_stats(frames, ranks) from the same launch; the histogram table is then scratch
            r = _device_ranks(ranks, x.device)
            st = torch.empty((n, r.numel()), dtype=torch.int32, device=x.device)
            check(_lib.load().pl_hist16_wl(x.data_ptr(), _dt(x), n, x.shape[1], x.shape[2], out.data_ptr(), tmax.data_ptr(),
                                           int(edge_window), emin.data_ptr(), emax.data_ptr(), r.data_ptr(), r.numel(), st.data_ptr(),
                                           _stream()), "pl_hist16_wl")
            return None, tmax, emin, emax, st
        check(_lib.load().pl_hist16_wl(x.data_ptr(), _dt(x), n, x.shape[1], x.shape[2], out.data_ptr(), tmax.data_ptr(), int(edge_window),
                                       emin.data_ptr(), emax.data_ptr(), 0, 0, 0, _stream()), "pl_hist16_wl")
        return out, tmax, emin, emax
    check(_lib.load().pl_hist16_tiles(x.data_ptr(), _dt(x), n, x[0].numel(), out.data_ptr(), tmax.data_ptr(), _stream()),
          "pl_hist16_tiles")
    return out, tmax


def otsu_from_hist(hist: torch.Tensor, dtype: torch.dtype):
    n = hist.shape[0]
    thr = torch.empty(n, dtype=torch.int32, device=hist.device)
    mn = torch.empty_like(thr)
    mx = torch.empty_like(thr)
    check(_lib.load().pl_otsu_from_hist(hist.data_ptr(), _DTYPES[dtype], n, thr.data_ptr(), mn.data_ptr(),
                                        mx.data_ptr(), _stream()), "pl_otsu_from_hist")
    return thr, mn, mx


def otsu16(frames: torch.Tensor, lo: torch.Tensor | None = None, hi: torch.Tensor | None = None,
           hist: torch.Tensor | None = None):
    """``skimage.filters.threshold_otsu`` per 16-bit frame -> (threshold, min, max) int32 [N]: one read of the frame with
    the histogram in an LDS window where its values span <= 38 912 bins, the two-kernel path otherwise (``pl_otsu16``).
    ``lo`` / ``hi``: optional per-frame bounds (int32 [N], ``lo <= values <= hi``) that place the window."""
    x = _frames(frames)
    if x.dtype not in (torch.uint16, torch.int16):
        raise TypeError("otsu16 needs uint16 or int16 frames")
    n, dev = x.shape[0], x.device
    thr = torch.empty(n, dtype=torch.int32, device=dev)
    mn, mx, flag = torch.empty_like(thr), torch.empty_like(thr), torch.empty_like(thr)
    hist = torch.empty((n, 65536), dtype=torch.int32, device=dev) if hist is None else hist
    check(_lib.load().pl_otsu16(x.data_ptr(), _dt(x), n, x[0].numel(), None if lo is None else lo.data_ptr(),
                                None if hi is None else hi.data_ptr(), thr.data_ptr(),
                                mn.data_ptr(), mx.data_ptr(), flag.data_ptr(), hist.data_ptr(), _stream()), "pl_otsu16")
    return thr, mn, mx


def median3_otsu16(frames: torch.Tensor, hist: torch.Tensor | None = None, scratch: torch.Tensor | None = None):
    """:func:`otsu16` of ``median_filter(frame, size=3)`` without writing the median plane (``pl_median3_otsu16``: the one-pass
    Otsu kernel computes the medians on the fly) -> (threshold, min, max, flag) int32 [N]; ``flag[i] = 1`` marks the frames that
    did not fit the one-pass window: their median plane is in ``scratch`` and they went through the two-kernel path.
    Frames need width % 8 == 0 and more than one row."""
    x = _frames(frames)
    if x.dtype not in (torch.uint16, torch.int16):
        raise TypeError("median3_otsu16 needs uint16 or int16 frames")
    n, h, w = x.shape
    dev = x.device
    thr = torch.empty(n, dtype=torch.int32, device=dev)
    mn, mx, flag = torch.empty_like(thr), torch.empty_like(thr), torch.empty_like(thr)
    hist = torch.empty((n, 65536), dtype=torch.int32, device=dev) if hist is None else hist
    scratch = torch.empty_like(x) if scratch is None else scratch
    check(_lib.load().pl_median3_otsu16(x.data_ptr(), scratch.data_ptr(), _dt(x), n, h, w, None, None, thr.data_ptr(),
                                        mn.data_ptr(), mx.data_ptr(), flag.data_ptr(), hist.data_ptr(), _stream()),
          "pl_median3_otsu16")
    return thr, mn, mx, flag


def median3_threshold_colsum_u16(frames: torch.Tensor, thr_i32: torch.Tensor, out=None, colsum=None):
    """``threshold(median_filter(frame, 3), thr)`` and its axis-0 column sums in one pass over the unfiltered frame
    (``pl_median3_threshold_colsum_u16``) -> (thresholded uint16 frames, int64 [N, W] column sums)."""
    x = _frames(frames)
    if x.dtype != torch.uint16:
        raise TypeError("median3_threshold_colsum_u16 needs uint16 frames")
    n, h, w = x.shape
    out = torch.empty_like(x) if out is None else out
    colsum = torch.empty((n, w), dtype=torch.int64, device=x.device) if colsum is None else colsum
    check(_lib.load().pl_median3_threshold_colsum_u16(x.data_ptr(), out.data_ptr(), n, h, w, thr_i32.data_ptr(),
                                                      colsum.data_ptr(), _stream()), "pl_median3_threshold_colsum_u16")
    return out, colsum


def threshold_otsu(frames: torch.Tensor) -> torch.Tensor:
    """``skimage.filters.threshold_otsu`` per integer frame -> int32 [N]."""
    return otsu16(frames)[0]


def _percentile_plan(cnt: int, q):
    """numpy's default 'linear' method: virtual index q/100*(n-1) -> (lower rank, upper rank, gamma)."""
    qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
    if np.any(qs < 0) or np.any(qs > 100):
        raise ValueError("Percentiles must be in the range [0, 100]")
    virt = (qs / 100.0) * (cnt - 1)
    lo = np.floor(virt).astype(np.int64)
    hi = np.minimum(lo + 1, cnt - 1)
    return qs, lo, hi, virt - lo


_RANK_CACHE: dict = {}


def _device_ranks(ranks, device) -> torch.Tensor:
    """the rank list on the device, cached: a blocking host-to-device copy would synchronise the stream on every call"""
    key = (tuple(int(r) for r in np.asarray(ranks).ravel()), str(device))
    t = _RANK_CACHE.get(key)
    if t is None:
        if len(_RANK_CACHE) > 64:
            _RANK_CACHE.clear()
        t = _RANK_CACHE[key] = torch.as_tensor(np.asarray(ranks, dtype=np.int64)).to(device)
    return t


def order_stats(frames: torch.Tensor, ranks, hist=None) -> torch.Tensor:
    """Exact order statistics (0-based ranks) of every 16-bit frame -> int32 [N, len(ranks)] on the
    device (no host synchronisation)."""
    x = _frames(frames)
    n = x.shape[0]
    r = _device_ranks(ranks, x.device)
    out = torch.empty((n, r.numel()), dtype=torch.int32, device=x.device)
    hist = histogram16(x) if hist is None else hist
    check(_lib.load().pl_order_stats_from_hist(hist.data_ptr(), _dt(x), n, r.data_ptr(), r.numel(),
                                               out.data_ptr(), _stream()), "pl_order_stats_from_hist")
    return out


def lerp_like_numpy(a: torch.Tensor, b: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """numpy ``_lerp`` on float64 tensors: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5."""
    d = b - a
    return torch.where(t >= 0.5, b - d * (1 - t), a + d * t)


def percentile(frames: torch.Tensor, q) -> torch.Tensor:
    """``np.percentile(frame, q)`` (default linear method) per 16-bit frame -> float64 [N, len(q)].
    The two neighbouring order statistics come from the exact device histogram; the final
    interpolation is numpy's ``_lerp`` formula evaluated in float64 on the host."""
    x = _frames(frames)
    cnt = x[0].numel()
    qs, lo, hi, frac = _percentile_plan(cnt, q)
    st = order_stats(x, np.concatenate([lo, hi])).cpu().numpy().astype(np.float64)
    a, b = st[:, : len(qs)], st[:, len(qs):]
    d = b - a
    res = a + d * frac
    res = np.where((frac >= 0.5)[None, :], b - d * (1 - frac), res)
    return torch.from_numpy(res)


# ------------------------------------------------------------------------------------- profiles
def reduce_axis(frames: torch.Tensor, axis: int, op: str = "mean") -> torch.Tensor:
    """``np.<op>(frame, axis)`` per frame -> float64 [N, W] (axis 0) or [N, H] (axis 1)."""
    x = _frames(frames)
    n, h, w = x.shape
    out = torch.empty((n, w if axis == 0 else h), dtype=torch.float64, device=x.device)
    check(_lib.load().pl_reduce_axis(x.data_ptr(), _dt(x), n, h, w, int(axis), _REDUCE[op], out.data_ptr(),
                                     _stream()), "pl_reduce_axis")
    return out


def threshold_colsum_u16(frames: torch.Tensor, thr_i32: torch.Tensor, out=None, colsum=None):
    x = _frames(frames)
    if x.dtype != torch.uint16:
        raise TypeError("threshold_colsum_u16 needs uint16 frames")
    n, h, w = x.shape
    out = torch.empty_like(x) if out is None else out
    colsum = torch.empty((n, w), dtype=torch.int64, device=x.device) if colsum is None else colsum
    check(_lib.load().pl_threshold_colsum_u16(x.data_ptr(), out.data_ptr(), n, h, w, thr_i32.data_ptr(),
                                              colsum.data_ptr(), _stream()), "pl_threshold_colsum_u16")
    return out, colsum


# ---------------------------------------------------------------------------------------- peaks
PEAK_PROP_KEYS = ("peak_heights", "prominences", "widths", "width_heights", "left_ips", "right_ips")


@dataclass
class PeakBatch:
    """Device-resident result of :func:`find_peaks_batch` (capacity ``cap`` per profile)."""

    count: torch.Tensor       # int32 [N]
    idx: torch.Tensor         # int32 [N, cap]
    left_bases: torch.Tensor  # int32 [N, cap]
    right_bases: torch.Tensor # int32 [N, cap]
    props: torch.Tensor       # float64 [N, 6, cap] in PEAK_PROP_KEYS order
    status: torch.Tensor      # int32 [N]: 0 ok, 1 truncated to cap, 2 too many candidate maxima

    def to_host(self, i: int = 0):
        """(peak_idxs, peak_props) for profile ``i`` in the reference's return format."""
        st = int(self.status[i])
        if st != 0:
            raise _lib.PylinacHipError(
                f"find_peaks: profile {i} overflowed the peak capacity (status {st}); raise `cap`")
        c = int(self.count[i])
        idx = self.idx[i, :c].cpu().numpy().astype(np.intp)
        p = self.props[i, :, :c].cpu().numpy()
        props = {k: p[j].copy() for j, k in enumerate(PEAK_PROP_KEYS)}
        props["left_bases"] = self.left_bases[i, :c].cpu().numpy().astype(np.intp)
        props["right_bases"] = self.right_bases[i, :c].cpu().numpy().astype(np.intp)
        return idx, props


def make_peak_params(length: int, threshold=-np.inf, peak_separation=0, max_number=None,
                     fwxm_height: float = 0.5, min_width=0, search_region=(0.0, 1.0),
                     peak_sort: str = "prominences", required_prominence=None) -> PeakParams:
    """``_parse_peak_args`` (pylinac/core/profile.py:2626-2649) minus the data-dependent part
    (the ratio threshold needs the profile's min/max and is resolved on the device)."""
    p = PeakParams()
    if 0 <= threshold <= 1:
        p.threshold_is_ratio = 1
    p.threshold = float(threshold)
    if 0 <= peak_separation <= 1:
        peak_separation = max(int(peak_separation * length), 1)
    p.distance = max(int(math.ceil(peak_separation)), 1)
    if max(search_region) <= 1:
        lo = int(search_region[0] * length)
        hi = int(search_region[1] * length)
    else:
        lo, hi = search_region[0], search_region[1]
    sl = range(length)[lo:hi]  # python slice semantics of values[lo:hi]
    p.region_lo = sl.start
    p.region_hi = sl.stop if len(sl) else sl.start
    p.has_prominence = 0 if required_prominence is None else 1
    p.prominence_min = 0.0 if required_prominence is None else float(required_prominence)
    p.width_min = float(min_width)
    p.rel_height = 1 - fwxm_height
    p.max_number = -1 if max_number is None else int(max_number)
    if peak_sort not in _lib.PL_SORT:
        raise KeyError(peak_sort)
    p.sort_key = _lib.PL_SORT[peak_sort]
    return p


def find_peaks_batch(profiles: torch.Tensor, cap: int | None = None, lens: torch.Tensor | None = None,
                     regions: torch.Tensor | None = None, **kwargs) -> PeakBatch:
    """``pylinac.core.profile.find_peaks`` for every row of ``profiles`` [N, L] (float64).
    ``lens`` (int32 [N]) makes the batch ragged: row i holds ``lens[i] <= L`` samples.
    ``regions`` (int32 [N, 2]): per-profile search region ``values[lo:hi]`` (indices; overrides ``search_region``)."""
    x = profiles
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if not x.is_cuda:
        raise ValueError("profiles must live on the GPU")
    if x.dtype != torch.float64:
        x = x.to(torch.float64)
    x = x.contiguous()
    n, length = x.shape
    if length == 0:
        raise ValueError("Array must not be empty")
    prm = make_peak_params(length, **kwargs)
    if cap is None:
        span = length if regions is not None else prm.region_hi - prm.region_lo
        cap = prm.max_number if prm.max_number > 0 else max(span // 2 + 1, 1)
        cap = max(cap, 1)
    dev = x.device
    if regions is not None:
        regions = regions.to(device=dev, dtype=torch.int32).contiguous()
        if tuple(regions.shape) != (n, 2):
            raise ValueError("regions must be [N, 2]")
    res = PeakBatch(
        count=torch.empty(n, dtype=torch.int32, device=dev),
        idx=torch.empty((n, cap), dtype=torch.int32, device=dev),
        left_bases=torch.empty((n, cap), dtype=torch.int32, device=dev),
        right_bases=torch.empty((n, cap), dtype=torch.int32, device=dev),
        props=torch.empty((n, 6, cap), dtype=torch.float64, device=dev),
        status=torch.empty(n, dtype=torch.int32, device=dev),
    )
    check(
        _lib.load().pl_find_peaks_regions(x.data_ptr(), n, length, 0 if lens is None else lens.data_ptr(), x.stride(0),
                                          C.byref(prm), 0 if regions is None else regions.data_ptr(), cap,
                                          res.count.data_ptr(), res.idx.data_ptr(), res.left_bases.data_ptr(),
                                          res.right_bases.data_ptr(), res.props.data_ptr(), res.status.data_ptr(),
                                          _stream()),
        "pl_find_peaks_regions",
    )
    return res


def peak_valley_regions(profiles: torch.Tensor, peak_kwargs: list, valley_kwargs: list):
    """``find_peaks`` and, between the outermost peaks found, ``find_valleys`` for every (profile, region) pair in one launch
    (``pl_peak_valley_regions``; CTP528CP504.mtf, pylinac/ct.py:1511-1544).  ``peak_kwargs[k]`` / ``valley_kwargs[k]``: the
    keyword arguments of ``find_peaks_batch`` for region k (``max_number`` required; the valleys' search region is set per
    profile by the peaks).  -> (peak counts int32 [N, R], peak heights float64 [N, R, cap_p], valley counts int32 [N, R],
    profile values at the valleys float64 [N, R, cap_v], means float64 [N, R, 2] = np.mean of the peak heights (NaN unless
    the region held exactly ``max_number`` peaks) and of the valley values (NaN when none)); NaN beyond a count."""
    x = profiles
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if not x.is_cuda:
        raise ValueError("profiles must live on the GPU")
    x = x.to(torch.float64).contiguous()
    n, length = x.shape
    r = len(peak_kwargs)
    if r != len(valley_kwargs) or not 1 <= r <= 16:
        raise ValueError("1..16 regions, one valley parameter set per peak parameter set")
    pk = (_lib.PeakParams * r)(*[make_peak_params(length, **kw) for kw in peak_kwargs])
    vl = (_lib.PeakParams * r)(*[make_peak_params(length, **kw) for kw in valley_kwargs])
    cap_p = max(max(p.max_number for p in pk), 1)
    cap_v = max(max(p.max_number for p in vl), 1)
    dev = x.device
    pc = torch.empty((n, r), dtype=torch.int32, device=dev)
    ph = torch.empty((n, r, cap_p), dtype=torch.float64, device=dev)
    vc = torch.empty((n, r), dtype=torch.int32, device=dev)
    vv = torch.empty((n, r, cap_v), dtype=torch.float64, device=dev)
    means = torch.empty((n, r, 2), dtype=torch.float64, device=dev)
    check(_lib.load().pl_peak_valley_regions(x.data_ptr(), n, length, x.stride(0), pk, vl, r, cap_p, cap_v, pc.data_ptr(),
                                             ph.data_ptr(), vc.data_ptr(), vv.data_ptr(), means.data_ptr(), _stream()),
          "pl_peak_valley_regions")
    return pc, ph, vc, vv, means


def hill_fit(x: torch.Tensor, y: torch.Tensor, lens: torch.Tensor | None = None, last_step: bool = False):
    """``Hill.fit`` (pylinac/core/hill.py:18-30 = ``scipy.optimize.curve_fit`` with the reference's start values) for a batch
    of windows on the device (``pl_hill_fit``: MINPACK's Levenberg-Marquardt restated, one lane per fit).  ``x``, ``y``
    float64 [N, M]; ``lens`` int32 [N] for ragged windows.  -> (params float64 [N, 4] = a, b, c, d; info int32 [N]: 1-4 =
    converged, what ``curve_fit`` accepts; nfev int32 [N])."""
    xs = x.to(torch.float64).contiguous()
    ys = y.to(torch.float64).contiguous()
    if xs.dim() != 2 or xs.shape != ys.shape:
        raise ValueError("x and y must be [N, M] of the same shape")
    if not xs.is_cuda:
        raise ValueError("the windows must live on the GPU")
    n, m = xs.shape
    if not 4 <= m <= 1024:
        raise ValueError("4 .. 1024 samples per fit")
    dev = xs.device
    work = torch.empty((8 * m, n), dtype=torch.float64, device=dev)
    params = torch.empty((n, 4), dtype=torch.float64, device=dev)
    info = torch.empty(n, dtype=torch.int32, device=dev)
    nfev = torch.empty(n, dtype=torch.int32, device=dev)
    if lens is not None:
        lens = lens.to(device=dev, dtype=torch.int32).contiguous()
    if last_step:           # + the relative length of the last accepted step (pl_hill_fit_ex): > ~1e-6 = stopped in a flat valley
        step = torch.empty(n, dtype=torch.float64, device=dev)
        check(_lib.load().pl_hill_fit_ex(xs.data_ptr(), ys.data_ptr(), 0 if lens is None else lens.data_ptr(), n, m, xs.stride(0),
                                         work.data_ptr(), params.data_ptr(), info.data_ptr(), nfev.data_ptr(), step.data_ptr(),
                                         _stream()), "pl_hill_fit_ex")
        return params, info, nfev, step
    check(_lib.load().pl_hill_fit(xs.data_ptr(), ys.data_ptr(), 0 if lens is None else lens.data_ptr(), n, m, xs.stride(0),
                                  work.data_ptr(), params.data_ptr(), info.data_ptr(), nfev.data_ptr(), _stream()), "pl_hill_fit")
    return params, info, nfev


def hill_windows(x_indices: torch.Tensor, values: torch.Tensor, peaks: PeakBatch, valleys: PeakBatch, window_ratio: float,
                 mmax: int):
    """The penumbra windows of ``SingleProfile.inflection_data`` (pylinac/core/profile.py:1676-1700) for processed profiles
    ``values`` float64 [N, S] sharing ``x_indices`` [S]; ``peaks`` / ``valleys`` = the derivative's extrema (index order).
    -> (xw, yw float64 [2N, mmax], lens int32 [2N], edges float64 [N, 2] = left_idx, right_idx)."""
    v = values.to(torch.float64).contiguous()
    xi = x_indices.to(device=v.device, dtype=torch.float64).contiguous()
    n, s = v.shape
    if xi.numel() != s:
        raise ValueError("x_indices and values must have the same length")
    dev = v.device
    xw = torch.zeros((2 * n, mmax), dtype=torch.float64, device=dev)
    yw = torch.zeros((2 * n, mmax), dtype=torch.float64, device=dev)
    lens = torch.empty(2 * n, dtype=torch.int32, device=dev)
    edges = torch.empty((n, 2), dtype=torch.float64, device=dev)
    check(_lib.load().pl_hill_windows(xi.data_ptr(), v.data_ptr(), n, s, peaks.count.data_ptr(), peaks.idx.data_ptr(),
                                      peaks.idx.shape[1], valleys.count.data_ptr(), valleys.idx.data_ptr(), valleys.idx.shape[1],
                                      float(window_ratio), int(mmax), xw.data_ptr(), yw.data_ptr(), lens.data_ptr(),
                                      edges.data_ptr(), _stream()), "pl_hill_windows")
    return xw, yw, lens, edges


def hill_inflection(params: torch.Tensor) -> torch.Tensor:
    """``Hill.inflection_idx`` and ``Hill.y`` there (pylinac/core/hill.py:32-36, 56-65): params [..., 4] -> [..., 2]."""
    p = params.to(torch.float64).contiguous()
    out = torch.empty(p.shape[:-1] + (2,), dtype=torch.float64, device=p.device)
    check(_lib.load().pl_hill_inflection(p.data_ptr(), p.numel() // 4, out.data_ptr(), _stream()), "pl_hill_inflection")
    return out


def hill_penumbra(params: torch.Tensor, inflection: torch.Tensor, lower: float, upper: float) -> torch.Tensor:
    """The Hill-method part of ``SingleProfile.penumbra`` (pylinac/core/profile.py:1852-1898): params [..., 4], inflection
    [..., 2] (index, value: :func:`hill_inflection`) -> [..., 6] = lower index, lower value, upper index, upper value, width,
    gradient at the inflection point."""
    p = params.to(torch.float64).contiguous()
    q = inflection.to(torch.float64).contiguous()
    out = torch.empty(p.shape[:-1] + (6,), dtype=torch.float64, device=p.device)
    check(_lib.load().pl_hill_penumbra(p.data_ptr(), q.data_ptr(), p.numel() // 4, float(lower), float(upper), out.data_ptr(),
                                       _stream()), "pl_hill_penumbra")
    return out


def profile_lookup(x_indices: torch.Tensor, values: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """``SingleProfile._y_original_to_interp`` per profile: values float64 [N, S], q float64 [N, Q] (or [N]) -> like q."""
    v = values.to(torch.float64).contiguous()
    xi = x_indices.to(device=v.device, dtype=torch.float64).contiguous()
    n, s = v.shape
    qq = q.to(torch.float64).reshape(n, -1).contiguous()
    out = torch.empty_like(qq)
    check(_lib.load().pl_profile_lookup(xi.data_ptr(), v.data_ptr(), n, s, qq.data_ptr(), qq.shape[1], out.data_ptr(),
                                        _stream()), "pl_profile_lookup")
    return out.reshape(q.shape)


def index_to_original(x_indices: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """``SingleProfile._x_interp_to_original``: ``np.interp(q, arange(S), x_indices)`` (what scipy's non-extrapolating linear
    ``interp1d`` delegates to) for positions q (any shape, float64) in sample coordinates."""
    xi = x_indices.to(torch.float64).contiguous()
    qq = q.to(device=xi.device, dtype=torch.float64).contiguous()
    out = torch.empty_like(qq)
    check(_lib.load().pl_index_to_original(xi.data_ptr(), xi.numel(), qq.data_ptr(), qq.numel(), out.data_ptr(), _stream()),
          "pl_index_to_original")
    return out


def fwxm_record(res: PeakBatch, out=None) -> torch.Tensor:
    """FWXMProfile edges / centre / width from a ``max_number=1`` peak batch -> float64 [N, 8]."""
    n = res.count.shape[0]
    cap = res.idx.shape[1]
    out = torch.empty((n, 8), dtype=torch.float64, device=res.count.device) if out is None else out
    check(_lib.load().pl_fwxm_record(res.count.data_ptr(), res.idx.data_ptr(), res.props.data_ptr(), cap, n,
                                     out.data_ptr(), _stream()), "pl_fwxm_record")
    return out


# ------------------------------------------------------------------------- circle profiles, Sobel
_circle_cache: dict = {}
CIRCLE_RING = os.environ.get("PL_CIRCLE_RING", "1") != "0"


def _circle_tables(size: float, start_angle: float, ccw: bool, dev):
    """cos / sin of ``circle_radians`` on the device, cached per ring geometry (numpy's libm values: see csrc/circle.hip)."""
    key = (float(size), float(start_angle), bool(ccw), str(dev))
    hit = _circle_cache.get(key)
    if hit is None:
        rads = circle_radians(size, start_angle, ccw)
        both = torch.from_numpy(np.stack([np.cos(rads), np.sin(rads)])).to(dev)
        if len(_circle_cache) > 64:
            _circle_cache.clear()
        hit = _circle_cache[key] = (both[0], both[1], len(rads))
    return hit


def circle_radians(size: float, start_angle: float = 0, ccw: bool = True) -> np.ndarray:
    """``CircleProfile._radians`` (pylinac/core/profile.py:2244-2252)."""
    interval = (2 * np.pi) / size
    rads = np.arange(0 + start_angle, (2 * np.pi) + start_angle - interval, interval)
    if ccw:
        rads = rads[::-1]
    return rads


_ring_cache: dict = {}


def _ring_args(sidx, radii: np.ndarray, n: int, dev):
    """Slice index [n] (int64) and radii [n, nr] on the device, cached: for a CatPhan pass they are the same numbers batch after
    batch, and a pageable host-to-device copy in the middle of a pass waits for everything queued before it."""
    r = np.ascontiguousarray(radii, dtype=np.float64)
    key = (None if sidx is None else sidx.tobytes(), r.tobytes(), r.shape, n, str(dev))
    hit = _ring_cache.get(key)
    if hit is None:
        if r.ndim == 1:
            r = np.broadcast_to(r[None, :], (n, r.shape[0]))
        pack = np.empty(n * (r.shape[1] + 1), dtype=np.float64)
        pack[:n].view(np.int64)[:] = sidx if sidx is not None else 0
        pack[n:] = r.reshape(-1)
        d = torch.from_numpy(pack).to(dev)
        if len(_ring_cache) > 16:
            _ring_cache.clear()
        hit = _ring_cache[key] = (d[:n].view(torch.int64), d[n:].view(n, r.shape[1]))
    return hit


def circle_profile(frames: torch.Tensor, cx, cy, radii, size: float, start_angle: float = 0,
                   ccw: bool = True, divisor: float = 1.0, combine=None, want_margin: bool = False):
    """``ndimage.map_coordinates(order=0)`` along circles, summed over ``radii`` and divided by
    ``divisor`` (pylinac/core/profile.py:2279-2283, 2473-2483).  ``cx, cy``: scalar or [N] (numpy / python numbers, or
    float64 DEVICE tensors [N]: nothing is uploaded then); ``radii``: [nr] or [N, nr]; ``size`` = pi * r_max * 2 *
    sampling_ratio.  -> float64 [N, nsamp].
    ``combine=(slice_index, slices_per_volume, plusminus)``: ``frames`` is a stack of whole volumes and profile i is taken
    on ``combine_surrounding_slices(slice_index[i] +- plusminus, "max")`` of its own volume (pylinac/ct.py:3351-3386) --
    the maximum is formed per tap, the combined slices are never built; N is then ``len(slice_index)``.
    ``want_margin`` (with ``combine``) -> (profiles, margin float64 [N]): how far each profile's centre may move before any
    of its samples changes (``pl_circle_profile_combined_ex``)."""
    x = _frames(frames)
    n_stack, h, w = x.shape
    dev = x.device
    if combine is not None:
        sidx, spv, pm = combine
        sidx = np.ascontiguousarray(sidx, dtype=np.int64)
        n = len(sidx)
    else:
        sidx = None
        n = n_stack
    d_cos, d_sin, nsamp = _circle_tables(size, start_angle, ccw, dev)
    r_host = np.asarray(radii, dtype=np.float64)
    if isinstance(cx, torch.Tensor) and isinstance(cy, torch.Tensor):
        d_sidx, r = _ring_args(sidx, r_host, n, dev)
        cxs, cys = cx.to(torch.float64).contiguous(), cy.to(torch.float64).contiguous()
        if cxs.numel() != n or cys.numel() != n or cxs.device != dev:
            raise ValueError("device centres: one (x, y) per profile, on the frames' device")
    else:
        # ONE host-to-device copy for the per-profile arguments (slice index, radii, centres): five separate pageable copies
        # were five synchronisations per call
        r = r_host
        if r.ndim == 1:
            r = np.broadcast_to(r[None, :], (n, r.shape[0]))
        nr = r.shape[1]
        pack = np.empty(n * (nr + 3), dtype=np.float64)
        pack[:n].view(np.int64)[:] = sidx if sidx is not None else 0
        pack[n:n + n * nr] = r.reshape(-1)
        pack[n + n * nr:2 * n + n * nr] = np.broadcast_to(np.asarray(cx, dtype=np.float64), (n,))
        pack[2 * n + n * nr:] = np.broadcast_to(np.asarray(cy, dtype=np.float64), (n,))
        d_pack = torch.from_numpy(pack).to(dev)
        d_sidx = d_pack[:n].view(torch.int64)
        r = d_pack[n:n + n * nr].view(n, nr)
        cxs = d_pack[n + n * nr:2 * n + n * nr]
        cys = d_pack[2 * n + n * nr:]
    out = torch.empty((n, nsamp), dtype=torch.float64, device=dev)
    if combine is not None:
        margin = torch.full((n,), float("inf"), dtype=torch.float64, device=dev) if want_margin else None
        # the radii are host numbers: their range is the promise that lets a thin ring be staged in LDS (csrc/circle.hip)
        r_abs = np.abs(r_host)
        r_lo, r_hi = (float(r_abs.min()), float(r_abs.max())) if r_abs.size and np.isfinite(r_abs).all() else (0.0, np.inf)
        if not CIRCLE_RING:                                  # A/B knob: an infinite promise forwards to the gather kernel
            r_lo, r_hi = 0.0, np.inf
        check(_lib.load().pl_circle_profile_ring(x.data_ptr(), _dt(x), n_stack, h, w, d_sidx.data_ptr(), n, int(spv),
                                                 int(pm), d_cos.data_ptr(), d_sin.data_ptr(), nsamp, r.data_ptr(),
                                                 r.shape[1], cxs.data_ptr(), cys.data_ptr(), float(divisor), r_lo, r_hi,
                                                 out.data_ptr(), 0 if margin is None else margin.data_ptr(),
                                                 _stream()), "pl_circle_profile_ring")
        return (out, margin) if want_margin else out
    if want_margin:
        raise ValueError("want_margin needs combine=")
    check(_lib.load().pl_circle_profile(x.data_ptr(), _dt(x), n, h, w, d_cos.data_ptr(), d_sin.data_ptr(),
                                        nsamp, r.data_ptr(), r.shape[1], cxs.data_ptr(), cys.data_ptr(),
                                        float(divisor), out.data_ptr(), _stream()), "pl_circle_profile")
    return out


def phantom_axis_fit(roi: torch.Tensor, n_volumes: int, x_adjustment: float = 0.0, y_adjustment: float = 0.0):
    """``pl_phantom_axis_fit``: the device's placement fit of ``CatPhanBase.find_phantom_axis`` (pylinac/ct.py:2398-2446) over
    the ROI tables [n_volumes * spv, 8] of ``edge_regions``.  -> (fit float64 [V, 4] = zx slope, zx intercept, zy slope, zy
    intercept; centres float64 [V * spv, 2] = (x, y); flag int32 [V])."""
    t = roi.contiguous()
    n = t.shape[0]
    if t.dim() != 2 or t.shape[1] != 8 or t.dtype != torch.float64 or n % int(n_volumes):
        raise ValueError("roi: float64 [n_volumes * slices_per_volume, 8]")
    spv = n // int(n_volumes)
    dev = t.device
    fit = torch.empty((int(n_volumes), 4), dtype=torch.float64, device=dev)
    centers = torch.empty((n, 2), dtype=torch.float64, device=dev)
    flag = torch.empty(int(n_volumes), dtype=torch.int32, device=dev)
    check(_lib.load().pl_phantom_axis_fit(t.data_ptr(), int(n_volumes), spv, float(x_adjustment), float(y_adjustment),
                                          fit.data_ptr(), centers.data_ptr(), flag.data_ptr(), _stream()), "pl_phantom_axis_fit")
    return fit, centers, flag


def sobel(frames: torch.Tensor, axis: int) -> torch.Tensor:
    """``scipy.ndimage.sobel(frame, axis)`` per frame (pylinac/core/image.py:1006-1007)."""
    x = _frames(frames)
    n, h, w = x.shape
    out = torch.empty_like(x)
    check(_lib.load().pl_sobel(x.data_ptr(), out.data_ptr(), _dt(x), n, h, w, int(axis), _stream()), "pl_sobel")
    return out


# ------------------------------------------------------------ components / fill holes / centroid
def _mask(m: torch.Tensor) -> torch.Tensor:
    if m.dtype == torch.bool:
        m = m.to(torch.uint8)
    x = _frames(m)
    if x.dtype != torch.uint8:
        raise TypeError("mask must be uint8 or bool")
    return x


def label(mask: torch.Tensor, connectivity: int = 4):
    """``skimage.measure.label`` numbering per frame -> (int32 labels [N,H,W], int32 count [N]).
    ``connectivity``: 4 (skimage connectivity=1) or 8 (skimage default in 2-D)."""
    x = _mask(mask)
    n, h, w = x.shape
    labels = torch.empty((n, h, w), dtype=torch.int32, device=x.device)
    work = torch.empty((n, h, w), dtype=torch.int32, device=x.device)
    count = torch.empty(n, dtype=torch.int32, device=x.device)
    check(_lib.load().pl_label(x.data_ptr(), n, h, w, int(connectivity), labels.data_ptr(), work.data_ptr(),
                               count.data_ptr(), _stream()), "pl_label")
    return labels, count


def fill_holes(mask: torch.Tensor, connectivity_bg: int = 4) -> torch.Tensor:
    """``scipy.ndimage.binary_fill_holes`` per frame (uint8 0/1)."""
    x = _mask(mask)
    n, h, w = x.shape
    out = torch.empty_like(x)
    work = torch.empty((n, h, w), dtype=torch.int32, device=x.device)
    flags = torch.empty((n, h, w), dtype=torch.uint8, device=x.device)
    check(_lib.load().pl_fill_holes(x.data_ptr(), out.data_ptr(), n, h, w, int(connectivity_bg), work.data_ptr(),
                                    flags.data_ptr(), _stream()), "pl_fill_holes")
    return out


def binary_centroid(mask: torch.Tensor) -> torch.Tensor:
    """``scipy.ndimage.center_of_mass`` of each mask -> float64 [N,3] = (row, col, count)."""
    x = _mask(mask)
    n, h, w = x.shape
    sums = torch.empty((n, 3), dtype=torch.int64, device=x.device)
    out = torch.empty((n, 3), dtype=torch.float64, device=x.device)
    check(_lib.load().pl_binary_centroid(x.data_ptr(), n, h, w, sums.data_ptr(), out.data_ptr(), _stream()),
          "pl_binary_centroid")
    return out


def field_cax(frames: torch.Tensor, sub, div, thr, defer: bool = False, tile_max: torch.Tensor | None = None):
    """``center_of_mass(binary_fill_holes(((a - sub) / div) >= thr))`` per frame -> float64 [N,3] = (row, col, count):
    the fused window path (``pl_field_cax``), the general mask -> fill -> centroid path for frames whose foreground
    bounding box does not fit the LDS window.  ``defer=True`` returns ``(out, status)`` without looking at the status on
    the host (no synchronisation): the caller redoes the frames whose status is non-zero.  ``tile_max``: see
    :func:`histogram16` -- same results, without a second full read of the frames."""
    x = _frames(frames)
    n, h, w = x.shape
    dev = x.device
    a = [_per_frame(v, n, dev)[0].expand(n).contiguous() if _per_frame(v, n, dev)[1] == 0 else _per_frame(v, n, dev)[0]
         for v in (sub, div, thr)]
    acc = torch.empty((n, 8), dtype=torch.int64, device=dev)
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    if tile_max is not None:          # histogram16(..., tiles=True)'s maxima: only tiles that can hold foreground are read
        if tile_max.shape != (n, (h * w + 511) // 512) or tile_max.dtype != torch.int16:
            raise ValueError("tile_max must be histogram16(frames, tiles=True)[1] of the same frames")
        check(_lib.load().pl_field_cax_tiles(x.data_ptr(), _dt(x), n, h, w, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(),
                                             tile_max.contiguous().data_ptr(), acc.data_ptr(), out.data_ptr(), status.data_ptr(),
                                             _stream()), "pl_field_cax_tiles")
    else:
        check(_lib.load().pl_field_cax(x.data_ptr(), _dt(x), n, h, w, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(),
                                       acc.data_ptr(), out.data_ptr(), status.data_ptr(), _stream()), "pl_field_cax")
    if defer:
        return out, status
    redo = torch.nonzero(status).flatten()
    if redo.numel():
        sel = x.view(torch.int16)[redo].view(torch.uint16) if x.dtype == torch.uint16 else x[redo]
        binary = scaled_binary(sel, a[0][redo], a[1][redo], a[2][redo])
        out[redo] = binary_centroid(fill_holes(binary, connectivity_bg=4))
    return out


def pack_columns(columns, n: int) -> torch.Tensor:
    """One float64 [n, K] table from K per-unit device columns in ONE launch (``pl_pack_columns``).  ``columns``: a list of
    ``tensor`` or ``(tensor, offset, add)`` -- the tensor float64 or int32, of shape [n] or [n, ...] (the column is
    ``tensor.reshape(n, -1)[:, offset] + add``)."""
    import ctypes as C

    k = len(columns)
    if not 1 <= k <= 16:
        raise ValueError("1..16 columns")
    ptrs, is32 = (C.c_void_p * k)(), (C.c_int * k)()
    strides, offsets, adds = (C.c_int64 * k)(), (C.c_int64 * k)(), (C.c_double * k)()
    keep, dev = [], None
    for j, col in enumerate(columns):
        t, off, add = (col, 0, 0.0) if isinstance(col, torch.Tensor) else col
        if t.dtype not in (torch.float64, torch.int32):
            raise TypeError("pack_columns takes float64 or int32 columns")
        t = t.contiguous()
        if t.shape[0] != n:
            raise ValueError("every column needs n rows")
        keep.append(t)
        dev = t.device
        ptrs[j], is32[j] = t.data_ptr(), 1 if t.dtype == torch.int32 else 0
        strides[j], offsets[j], adds[j] = (t.numel() // n if n else 1), int(off), float(add)
    out = torch.empty((n, k), dtype=torch.float64, device=dev)
    check(_lib.load().pl_pack_columns(ptrs, is32, strides, offsets, adds, k, n, out.data_ptr(), _stream()), "pl_pack_columns")
    return out


def wl_decisions(stats: torch.Tensor, edge_min: torch.Tensor, edge_max: torch.Tensor, frac) -> dict:
    """``pl_wl_decisions``: the inversion check, the edge test and the field threshold of ``WLBaseImage.analyze`` from the
    order statistics table ``stats`` int32 [N,16] (min, max, lower / upper neighbours of the percentiles 5, 99.9, 0.01, 50,
    99.99, 5, 99.5), on the device -> dict(inverted, noisy int32 [N]; vmin, vmax, gmax, thr float64 [N])."""
    n, dev = stats.shape[0], stats.device
    frac = np.ascontiguousarray(frac, dtype=np.float64)
    if stats.shape[1] != 16 or frac.size != 7:
        raise ValueError("stats must be [N,16] and frac must hold 7 weights")
    i32 = lambda: torch.empty(n, dtype=torch.int32, device=dev)
    f64 = lambda: torch.empty(n, dtype=torch.float64, device=dev)
    out = dict(inverted=i32(), noisy=i32(), vmin=f64(), vmax=f64(), gmax=f64(), thr=f64())
    check(_lib.load().pl_wl_decisions(stats.contiguous().data_ptr(), edge_min.data_ptr(), edge_max.data_ptr(), n, frac.ctypes.data,
                                      out["inverted"].data_ptr(), out["noisy"].data_ptr(), out["vmin"].data_ptr(),
                                      out["vmax"].data_ptr(), out["gmax"].data_ptr(), out["thr"].data_ptr(), _stream()),
          "pl_wl_decisions")
    return out


def edge_minmax(frames: torch.Tensor, window: int = 2):
    """min / max over the four ``window``-wide edge strips of every 16-bit frame -> (int32 [N], int32 [N])."""
    x = _frames(frames)
    n, h, w = x.shape
    mn = torch.empty(n, dtype=torch.int32, device=x.device)
    mx = torch.empty_like(mn)
    check(_lib.load().pl_edge_minmax(x.data_ptr(), _dt(x), n, h, w, int(window), mn.data_ptr(), mx.data_ptr(), _stream()),
          "pl_edge_minmax")
    return mn, mx


def scaled_binary(frames: torch.Tensor, sub, div, thr) -> torch.Tensor:
    """``((a - sub) / div) >= thr`` in float64 per frame -> uint8 mask."""
    x = _frames(frames)
    n = x.shape[0]
    dev = x.device
    a = [_per_frame(v, n, dev)[0].expand(n).contiguous() if _per_frame(v, n, dev)[1] == 0 else _per_frame(v, n, dev)[0]
         for v in (sub, div, thr)]
    out = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    check(_lib.load().pl_scaled_binary(x.data_ptr(), _dt(x), n, x[0].numel(), a[0].data_ptr(), a[1].data_ptr(),
                                       a[2].data_ptr(), out.data_ptr(), _stream()), "pl_scaled_binary")
    return out


# ----------------------------------------------------------------- CatPhan localisation primitives
def scharr(frames: torch.Tensor) -> torch.Tensor:
    """``skimage.filters.scharr(frame.astype(float))`` -> float64 [N,H,W] (pylinac/ct.py:391,3327)."""
    x = _frames(frames)
    n, h, w = x.shape
    out = torch.empty((n, h, w), dtype=torch.float64, device=x.device)
    check(_lib.load().pl_scharr(x.data_ptr(), out.data_ptr(), _dt(x), n, h, w, _stream()), "pl_scharr")
    return out


def gaussian_filter_mode(frames: torch.Tensor, sigma: float, mode: str = "nearest") -> torch.Tensor:
    """``ndimage.gaussian_filter(frame, sigma, mode=...)``; ``skimage.filters.gaussian`` uses 'nearest'."""
    if mode == "reflect":
        return gaussian_filter(frames, sigma)       # the default entry hands the library the cached host taps as well
    x = _frames(frames)
    n, h, w = x.shape
    wts, _, lw = _device_weights(sigma, x.device)
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    check(_lib.load().pl_gaussian2d_mode(x.data_ptr(), out.data_ptr(), tmp.data_ptr(), _dt(x), n, h, w,
                                         wts.data_ptr(), lw, {"reflect": 0, "nearest": 1, "constant": 2}[mode], _stream()),
          "pl_gaussian2d_mode")
    return out


def minmax_masked(frames: torch.Tensor, mask: torch.Tensor):
    """min/max of ``frame[mask]`` for a mask shared by all frames (float64 frames)."""
    x = _frames(frames)
    if x.dtype != torch.float64:
        raise TypeError("minmax_masked needs float64 frames")
    n = x.shape[0]
    mn = torch.empty(n, dtype=torch.float64, device=x.device)
    mx = torch.empty(n, dtype=torch.float64, device=x.device)
    check(_lib.load().pl_minmax_masked(x.data_ptr(), mask.contiguous().data_ptr(), n, x[0].numel(), mn.data_ptr(),
                                       mx.data_ptr(), _stream()), "pl_minmax_masked")
    return mn, mx


def clip(frames: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    x = _frames(frames)
    out = torch.empty_like(x)
    check(_lib.load().pl_clip(x.data_ptr(), out.data_ptr(), _dt(x), x.shape[0], x[0].numel(), float(lo), float(hi),
                              _stream()), "pl_clip")
    return out


_OPS = {">=": 0, ">": 1, "<=": 2, "<": 3}


def compare(frames: torch.Tensor, thr, op: str = ">") -> torch.Tensor:
    x = _frames(frames)
    n = x.shape[0]
    t, stride = _per_frame(thr, n, x.device)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.load().pl_compare(x.data_ptr(), _dt(x), n, x[0].numel(), t.data_ptr(), stride, _OPS[op],
                                 out.data_ptr(), _stream()), "pl_compare")
    return out


def hist_uniform(frames: torch.Tensor, edges: torch.Tensor, mask: torch.Tensor | None = None) -> torch.Tensor:
    """``np.histogram(frame[mask], bins=edges)`` for uniform float64 ``edges`` [N, nbins+1] -> int32 [N, nbins]."""
    x = _frames(frames)
    if x.dtype != torch.float64:
        raise TypeError("hist_uniform needs float64 frames")
    n = x.shape[0]
    nb = edges.shape[1] - 1
    out = torch.empty((n, nb), dtype=torch.int32, device=x.device)
    check(_lib.load().pl_hist_uniform(x.data_ptr(), 0 if mask is None else mask.data_ptr(), n, x[0].numel(),
                                      edges.contiguous().data_ptr(), nb, out.data_ptr(), _stream()), "pl_hist_uniform")
    return out


def scharr_gaussian(frames: torch.Tensor, sigma: float, mask: torch.Tensor | None = None):
    """``skimage.filters.gaussian(skimage.filters.scharr(frame.astype(float)), sigma)`` (mode 'nearest') in one pass over
    int16 / uint16 frames (``pl_scharr_gaussian``; pylinac/ct.py:391, 3327-3328) -> (edges float64 [N,H,W], the maximum of
    the raw Scharr magnitude [N], min and max [N] of ``edges[mask]`` for a frame-shared uint8 mask)."""
    x = _frames(frames)
    n, h, w = x.shape
    dev = x.device
    wts, _, lw = _device_weights(sigma, dev)
    out = torch.empty((n, h, w), dtype=torch.float64, device=dev)
    rawmax = torch.empty(n, dtype=torch.float64, device=dev)
    lo = torch.empty(n, dtype=torch.float64, device=dev)
    hi = torch.empty(n, dtype=torch.float64, device=dev)
    check(_lib.load().pl_scharr_gaussian(x.data_ptr(), _dt(x), n, h, w, wts.data_ptr(), lw,
                                         0 if mask is None else mask.contiguous().data_ptr(), out.data_ptr(),
                                         rawmax.data_ptr(), lo.data_ptr(), hi.data_ptr(), _stream()), "pl_scharr_gaussian")
    return out, rawmax, lo, hi


def edge_plane(frames: torch.Tensor, sigma: float, spans: torch.Tensor | None = None, mask: torch.Tensor | None = None,
               dtype=torch.float32, want_plane: bool = True):
    """The streaming form of ``scharr_gaussian`` (``pl_edge_plane``): the smoothed Scharr plane as float32 (the float64 value
    rounded to nearest; ``dtype=torch.float64`` for the exact plane; ``want_plane=False``: extrema only) and the exact float64
    extrema over the pixels selected by ``spans`` (int32 [H, 2] column intervals per row, see ``row_spans``) or ``mask``
    (uint8 [H, W]).  -> (plane or None, raw Scharr maximum [N], min [N], max [N])."""
    x = _frames(frames)
    n, h, w = x.shape
    dev = x.device
    wts, _, lw = _device_weights(sigma, dev)
    if dtype not in (torch.float32, torch.float64):
        raise TypeError("edge_plane writes float32 or float64")
    out = torch.empty((n, h, w), dtype=dtype, device=dev) if want_plane else None
    rawmax = torch.empty(n, dtype=torch.float64, device=dev)
    lo = torch.empty(n, dtype=torch.float64, device=dev)
    hi = torch.empty(n, dtype=torch.float64, device=dev)
    if spans is not None:
        spans = spans.to(torch.int32).contiguous()
        if tuple(spans.shape) != (h, 2):
            raise ValueError("spans must be [H, 2]")
    check(_lib.load().pl_edge_plane(x.data_ptr(), _dt(x), n, h, w, wts.data_ptr(), lw,
                                    0 if spans is None else spans.data_ptr(), 0 if mask is None else mask.contiguous().data_ptr(),
                                    0 if out is None else out.data_ptr(), _lib.PL_F32 if dtype == torch.float32 else _lib.PL_F64,
                                    rawmax.data_ptr(), lo.data_ptr(), hi.data_ptr(), _stream()), "pl_edge_plane")
    return out, rawmax, lo, hi


def edge_plane32(frames: torch.Tensor, sigma: float, spans: torch.Tensor | None = None):
    """``edge_plane`` in packed float32 (``pl_edge_plane32``): the plane lies within ``bracket`` float32 bit patterns of the
    exact value instead of being its rounded image -- hand ``bracket`` to ``edge_otsu`` / ``edge_regions``, which then decide
    from it exactly like from the exact plane.  The extrema are the EXACT float64 ones (recomputed from candidates).
    -> (plane float32, raw maximum [N], min [N], max [N], status int32 [N] (1 = repeat the slice with ``edge_plane``), bracket)."""
    x = _frames(frames)
    n, h, w = x.shape
    dev = x.device
    if w % 2:
        raise ValueError("edge_plane32 needs an even width")
    wts, _, lw = _device_weights(sigma, dev)
    lib = _lib.load()
    out = torch.empty((n, h, w), dtype=torch.float32, device=dev)
    work = torch.empty(int(lib.pl_edge_plane32_work_bytes(n)) + 16, dtype=torch.uint8, device=dev)
    rawmax = torch.empty(n, dtype=torch.float64, device=dev)
    lo = torch.empty(n, dtype=torch.float64, device=dev)
    hi = torch.empty(n, dtype=torch.float64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    if spans is not None:
        spans = spans.to(torch.int32).contiguous()
        if tuple(spans.shape) != (h, 2):
            raise ValueError("spans must be [H, 2]")
    check(lib.pl_edge_plane32(x.data_ptr(), _dt(x), n, h, w, wts.data_ptr(), lw, 0 if spans is None else spans.data_ptr(),
                              out.data_ptr(), work.data_ptr(), rawmax.data_ptr(), lo.data_ptr(), hi.data_ptr(), status.data_ptr(),
                              _stream()), "pl_edge_plane32")
    return out, rawmax, lo, hi, status, int(lib.pl_edge_plane32_bracket())


def edge_otsu(plane: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor, frames: torch.Tensor | None = None, sigma: float = 1.0,
              spans: torch.Tensor | None = None, mask: torch.Tensor | None = None, scale: float = 1.0, return_work: bool = False,
              bracket: int = 1):
    """``threshold_otsu(plane[selection]) * scale`` for the plane of ``edge_plane`` in one launch (``pl_edge_otsu``;
    pylinac/ct.py:3334-3340).  ``lo`` / ``hi``: the selection's exact extrema (``edge_plane``'s); a float32 plane also needs the
    ``frames`` and ``sigma`` it was made from.  -> (threshold * scale, threshold) float64 [N][, int32 [N, 258]: the histogram,
    an internal counter, the number of pixels recomputed exactly]."""
    p = _frames(plane)
    n, h, w = p.shape
    dev = p.device
    if p.dtype == torch.float32:
        if frames is None:
            raise ValueError("a float32 plane needs the frames it was made from")
        x = _frames(frames)
        if tuple(x.shape) != (n, h, w):
            raise ValueError("frames and plane differ in shape")
        wts, _, lw = _device_weights(sigma, dev)
        raw_ptr, raw_dt, wp = x.data_ptr(), _dt(x), wts.data_ptr()
    elif p.dtype == torch.float64:
        raw_ptr, raw_dt, wp, lw = 0, _lib.PL_I16, 0, 0
    else:
        raise TypeError("edge_otsu needs a float32 or float64 plane")
    if spans is not None:
        spans = spans.to(torch.int32).contiguous()
    work = torch.empty((n, 258), dtype=torch.int32, device=dev)
    thr = torch.empty(n, dtype=torch.float64, device=dev)
    raw = torch.empty(n, dtype=torch.float64, device=dev)
    check(_lib.load().pl_edge_otsu_ex(p.data_ptr(), _dt(p), raw_ptr, raw_dt, n, h, w, wp, lw,
                                      0 if spans is None else spans.data_ptr(), 0 if mask is None else mask.contiguous().data_ptr(),
                                      lo.contiguous().data_ptr(), hi.contiguous().data_ptr(), float(scale), work.data_ptr(),
                                      thr.data_ptr(), raw.data_ptr(), int(bracket), _stream()), "pl_edge_otsu_ex")
    return (thr, raw, work) if return_work else (thr, raw)


def otsu_float_masked(frames: torch.Tensor, mask: torch.Tensor | None, scale: float = 1.0, lohi=None):
    """``skimage.filters.threshold_otsu(frame[mask])`` for float64 frames (256 bins over the min .. max of the selected
    pixels; pylinac/ct.py:3323, 3338-3340) entirely on the device -> (threshold * scale, threshold) float64 [N].
    ``lohi``: the selection's (min, max) when the caller already has them."""
    x = _frames(frames)
    if x.dtype != torch.float64:
        raise TypeError("otsu_float_masked needs float64 frames")
    n = x.shape[0]
    dev = x.device
    lib, st = _lib.load(), _stream()
    if lohi is not None:
        lo, hi = lohi
    elif mask is None:
        lo, hi = minmax(x)
    else:
        lo, hi = minmax_masked(x, mask)
    edges = torch.empty((n, 257), dtype=torch.float64, device=dev)
    check(lib.pl_linspace_edges(lo.data_ptr(), hi.data_ptr(), 256, n, edges.data_ptr(), st), "pl_linspace_edges")
    counts = hist_uniform(x, edges, mask)
    thr = torch.empty(n, dtype=torch.float64, device=dev)
    raw = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib.pl_otsu_from_counts(counts.data_ptr(), edges.data_ptr(), 256, n, float(scale), thr.data_ptr(), raw.data_ptr(),
                                  st), "pl_otsu_from_counts")
    return thr, raw


def combine_slices(stack: torch.Tensor, plusminus: int, mode: str = "max", slices_per_volume: int | None = None) -> torch.Tensor:
    """``combine_surrounding_slices`` (pylinac/ct.py:3351-3386) for every slice of ``stack`` [S, H, W]: "max" keeps the
    dtype, "mean" gives float64; the window z - k .. z + k indexes the slice's own volume (``slices_per_volume``
    consecutive slices; default: the whole stack is one volume) like the reference's Python list: negative indices wrap
    around, slices whose window passes the end of the volume (IndexError in the reference) are for the caller to discard."""
    x = _frames(stack)
    if mode not in ("max", "mean"):
        raise ValueError("mode must be 'max' or 'mean'")
    out = torch.empty_like(x) if mode == "max" else torch.empty(x.shape, dtype=torch.float64, device=x.device)
    check(_lib.load().pl_combine_slices(x.data_ptr(), out.data_ptr(), _dt(x), x.shape[0], x[0].numel(), int(plusminus),
                                        0 if mode == "max" else 1, int(slices_per_volume or x.shape[0]), _stream()),
          "pl_combine_slices")
    return out


def clear_border(mask: torch.Tensor, buffer_size: int = 0) -> torch.Tensor:
    """``skimage.segmentation.clear_border(bw, buffer_size)`` per frame (uint8 0/1)."""
    x = _mask(mask)
    n, h, w = x.shape
    out = torch.empty_like(x)
    work = torch.empty((n, h, w), dtype=torch.int32, device=x.device)
    flags = torch.empty((n, h, w), dtype=torch.uint8, device=x.device)
    check(_lib.load().pl_clear_border(x.data_ptr(), out.data_ptr(), n, h, w, int(buffer_size), work.data_ptr(),
                                      flags.data_ptr(), _stream()), "pl_clear_border")
    return out


REGION_FIELDS = ("area", "bbox_r0", "bbox_c0", "bbox_r1", "bbox_c1", "sum_r", "sum_c", "sum_w", "sum_wr", "sum_wc")


def region_stats(labels: torch.Tensor, intensity: torch.Tensor | None, max_labels: int):
    """Raw regionprops sums per label -> (float64 [N, max_labels, 10] in REGION_FIELDS order,
    int32 overflow [N])."""
    n, h, w = labels.shape
    dev = labels.device
    isum = torch.empty((n, max_labels, 7), dtype=torch.int64, device=dev)
    wsum = torch.empty((n, max_labels, 3), dtype=torch.float64, device=dev)
    stats = torch.empty((n, max_labels, 10), dtype=torch.float64, device=dev)
    ovf = torch.empty(n, dtype=torch.int32, device=dev)
    check(_lib.load().pl_region_stats(labels.contiguous().data_ptr(), 0 if intensity is None else intensity.data_ptr(),
                                      n, h, w, int(max_labels), isum.data_ptr(), wsum.data_ptr(), stats.data_ptr(),
                                      ovf.data_ptr(), _stream()), "pl_region_stats")
    return stats, ovf


MASK_REGION_FIELDS = REGION_FIELDS[:7]


def mask_regions_fits(h: int, w: int, max_labels: int) -> bool:
    """Whether ``mask_regions`` can hold an h x w frame (bit plane + run list in LDS) and a table of ``max_labels`` rows."""
    return bool(_lib.load().pl_mask_regions_fits(int(h), int(w), int(max_labels)))


def mask_regions(frames: torch.Tensor, thr=None, clear_border_ext: int = 0, fill_holes: bool = False, max_labels: int = 64,
                 return_mask: bool = False):
    """``regionprops(label(binary_fill_holes(clear_border(frame > thr))))`` of every frame in one launch
    (``pl_mask_regions``: a workgroup per frame, bit plane and row runs in LDS).  ``frames`` float64 with per-frame ``thr``
    (device float64 [N]) or a uint8 / bool mask with ``thr=None``; ``clear_border_ext`` = ``buffer_size + 1`` or 0 for no
    clearing.  -> (float64 [N, max_labels, 7] in ``MASK_REGION_FIELDS`` order, int32 count [N], int32 status [N]
    (1 = too many row runs: use the separate entry points for that frame)[, uint8 final mask])."""
    x = _frames(frames) if frames.dtype != torch.bool else _mask(frames)
    n, h, w = x.shape
    dev = x.device
    if x.dtype == torch.float64:
        if thr is None:
            raise ValueError("float64 frames need thresholds")
        t = thr.to(torch.float64).reshape(-1)
        if t.numel() == 1 and n != 1:
            t = t.expand(n)
        t = t.contiguous()
        if t.numel() != n:
            raise ValueError("one threshold per frame")
        tp = t.data_ptr()
    elif x.dtype == torch.uint8:
        if thr is not None:
            raise ValueError("a mask takes no threshold")
        tp = 0
    else:
        raise TypeError("mask_regions needs float64 frames or uint8 masks")
    table = torch.empty((n, int(max_labels), 7), dtype=torch.float64, device=dev)
    count = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    om = torch.empty((n, h, w), dtype=torch.uint8, device=dev) if return_mask else None
    check(_lib.load().pl_mask_regions(x.data_ptr(), _dt(x), tp, n, h, w, int(clear_border_ext), int(bool(fill_holes)),
                                      int(max_labels), table.data_ptr(), count.data_ptr(), status.data_ptr(),
                                      0 if om is None else om.data_ptr(), _stream()), "pl_mask_regions")
    return (table, count, status, om) if return_mask else (table, count, status)


def edge_regions(plane: torch.Tensor, frames: torch.Tensor, sigma: float, thr: torch.Tensor, clear_border_ext: int = 0,
                 fill_holes: bool = False, max_labels: int = 64, catphan_size: float | None = None,
                 rawmax: torch.Tensor | None = None, want_table: bool = True, return_mask: bool = False, bracket: int = 1):
    """``mask_regions`` on the float32 plane of ``edge_plane`` (``pl_edge_regions``): pixels the float32 value cannot decide
    against the threshold are recomputed exactly from ``frames``.  With ``catphan_size`` and ``rawmax`` the phantom ROI of
    ``Slice.phantom_roi`` (pylinac/ct.py:381-425) is chosen in the same launch.
    -> dict(table float64 [N, max_labels, 7] or None, count, status int32 [N], roi float64 [N, 8] or None, mask or None)."""
    p = _frames(plane)
    x = _frames(frames)
    if p.dtype != torch.float32:
        raise TypeError("edge_regions needs the float32 plane of edge_plane")
    n, h, w = p.shape
    if tuple(x.shape) != (n, h, w):
        raise ValueError("frames and plane differ in shape")
    dev = p.device
    wts, _, lw = _device_weights(sigma, dev)
    t = thr.to(torch.float64).reshape(-1).contiguous()
    if t.numel() != n:
        raise ValueError("one threshold per frame")
    table = torch.empty((n, int(max_labels), 7), dtype=torch.float64, device=dev) if want_table else None
    count = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    roi = torch.empty((n, 8), dtype=torch.float64, device=dev) if catphan_size is not None else None
    if roi is not None and rawmax is None:
        raise ValueError("the ROI selection needs rawmax")
    om = torch.empty((n, h, w), dtype=torch.uint8, device=dev) if return_mask else None
    check(_lib.load().pl_edge_regions_ex(p.data_ptr(), x.data_ptr(), _dt(x), wts.data_ptr(), lw, t.data_ptr(), n, h, w,
                                         int(clear_border_ext), int(bool(fill_holes)), int(max_labels),
                                         0 if table is None else table.data_ptr(), count.data_ptr(), status.data_ptr(),
                                         0 if om is None else om.data_ptr(), float(catphan_size or 0.0),
                                         0 if rawmax is None else rawmax.contiguous().data_ptr(),
                                         0 if roi is None else roi.data_ptr(), int(bracket), _stream()), "pl_edge_regions_ex")
    return dict(table=table, count=count, status=status, roi=roi, mask=om)


def region_moments(labels: torch.Tensor, max_labels: int):
    """Exact raw moments per label -> (int64 [N, max_labels, 6] = m00, m10 (sum r), m01 (sum c), m20, m02, m11 in image
    coordinates, int32 overflow [N]).  See ``regionprops.inertia_from_raw_moments`` for what is formed from them."""
    lab = labels if labels.ndim == 3 else labels[None]
    n, h, w = lab.shape
    mom = torch.empty((n, int(max_labels), 6), dtype=torch.int64, device=lab.device)
    ovf = torch.empty(n, dtype=torch.int32, device=lab.device)
    check(_lib.load().pl_region_moments(lab.contiguous().data_ptr(), n, h, w, int(max_labels), mom.data_ptr(),
                                        ovf.data_ptr(), _stream()), "pl_region_moments")
    return mom, ovf


def interp1d(x: torch.Tensor, y: torch.Tensor, xq: torch.Tensor, kind: str = "linear") -> torch.Tensor:
    """``scipy.interpolate.interp1d(x, y, kind, bounds_error=False, fill_value="extrapolate")(xq)`` for a batch
    of profiles: ``x`` float64 [L] (shared) or [N, L], ``y`` float64 [N, L] (or [L]), ``xq`` float64 [S]
    -> float64 [N, S] (pylinac/core/profile.py:1349-1358)."""
    if kind not in ("linear", "cubic"):
        raise ValueError("kind must be 'linear' or 'cubic'")
    y2 = (y if y.ndim == 2 else y[None]).to(torch.float64).contiguous()
    n, length = y2.shape
    xs = x.to(torch.float64).contiguous()
    if xs.shape[-1] != length or (xs.ndim == 2 and xs.shape[0] != n):
        raise ValueError("x and y must have the same length")
    stride = 0 if xs.ndim == 1 else length
    q = xq.to(torch.float64).contiguous()
    out = torch.empty((n, q.numel()), dtype=torch.float64, device=y2.device)
    work = torch.empty(3 * n * length, dtype=torch.float64, device=y2.device) if kind == "cubic" else None
    check(_lib.load().pl_interp1d(xs.data_ptr(), stride, y2.data_ptr(), n, length, q.data_ptr(), q.numel(),
                                  0 if kind == "linear" else 1, work.data_ptr() if work is not None else None,
                                  out.data_ptr(), _stream()), "pl_interp1d")
    return out if y.ndim == 2 else out[0]


def cubic_spline_moments(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Second derivatives at the knots of the not-a-knot cubic spline through (x, y) -- what
    ``interp1d(x, y, kind="cubic")`` interpolates with -- for ``x`` float64 [L] and ``y`` float64 [N, L] (or [L])
    -> float64 [N, L].  The tridiagonal solves run on the device (``pl_interp1d``'s workspace, see the header)."""
    yy = y if y.dim() == 2 else y.unsqueeze(0)
    yy = yy.to(torch.float64).contiguous()
    xx = x.to(device=yy.device, dtype=torch.float64).contiguous()
    n, length = yy.shape
    if xx.dim() != 1 or xx.numel() != length:
        raise ValueError("x must be [L] (shared abscissae)")
    work = torch.empty(3 * n * length, dtype=torch.float64, device=yy.device)
    xq = xx[:1].clone()
    out = torch.empty((n, 1), dtype=torch.float64, device=yy.device)
    check(_lib.load().pl_interp1d(xx.data_ptr(), 0, yy.data_ptr(), n, length, xq.data_ptr(), 1, 1, work.data_ptr(),
                                  out.data_ptr(), _stream()), "pl_interp1d")
    return work[: n * length].view(n, length).clone()


def zoom1d_cubic(y: torch.Tensor, zoom: float, grid_mode: bool = False) -> torch.Tensor:
    """``scipy.ndimage.zoom(profile, zoom, order=3, mode="nearest", grid_mode=grid_mode)`` for float64 profiles [N, L]
    (or [L]) -> float64 [N, round(L * zoom)] (pylinac/core/profile.py:370-376, 985-991)."""
    yy = y if y.dim() == 2 else y.unsqueeze(0)
    yy = yy.to(torch.float64).contiguous()
    n, length = yy.shape
    out_length = int(round(length * zoom))
    if out_length < 1:
        raise ValueError("zoom leaves no samples")
    work = torch.empty(n * (length + 24), dtype=torch.float64, device=yy.device)
    out = torch.empty((n, out_length), dtype=torch.float64, device=yy.device)
    check(_lib.load().pl_zoom1d_cubic(yy.data_ptr(), n, length, out_length, 1 if grid_mode else 0, work.data_ptr(),
                                      out.data_ptr(), _stream()),
          "pl_zoom1d_cubic")
    return out if y.dim() == 2 else out[0]


def gradient1d(y: torch.Tensor) -> torch.Tensor:
    """``np.gradient(y)`` (unit spacing) for [L] or [N, L] float64 profiles."""
    y2 = (y if y.ndim == 2 else y[None]).to(torch.float64).contiguous()
    out = torch.empty_like(y2)
    check(_lib.load().pl_gradient1d(y2.data_ptr(), y2.shape[0], y2.shape[1], out.data_ptr(), _stream()), "pl_gradient1d")
    return out if y.ndim == 2 else out[0]


# ---- BaseImage.rotate (pylinac/core/image.py:780-783 -> skimage.transform.rotate) -----------------------------------
def rotation_matrix(rows: int, cols: int, angle: float, center=None):
    """The 3x3 inverse map skimage.transform.rotate hands to warp(): translate the centre to the origin, rotate by
    ``angle`` degrees, translate back -- built with the same numpy expressions (SimilarityTransform parameter matrices,
    ``tform3 + tform2 + tform1`` == ``T1 @ (R @ T3)``, last row forced to (0, 0, 1))."""
    import math

    import numpy as np

    center = np.array((cols, rows)) / 2.0 - 0.5 if center is None else np.asarray(center)

    def similarity(rotation=0.0, translation=(0, 0)):
        m = np.array([[math.cos(rotation), -math.sin(rotation), 0], [math.sin(rotation), math.cos(rotation), 0], [0, 0, 1]])
        m[0:2, 0:2] *= 1
        m[0:2, 2] = translation
        return m

    t1 = similarity(translation=center)
    t2 = similarity(rotation=np.deg2rad(angle))
    t3 = similarity(translation=-center)
    m = t1 @ (t2 @ t3)
    m[2] = (0, 0, 1)
    return m


def warp_affine(frames: torch.Tensor, matrix, lo=None, hi=None, order: int = 1) -> torch.Tensor:
    """Order-1 (bilinear; ``order=0``: nearest neighbour, never clipped) 'edge' warp of float32 / float64 frames through the affine inverse map ``matrix`` (3x3 or 2x3, output
    (row, col) -> input (r, c)); the result is clipped to ``[lo, hi]`` per frame (default: each frame's own min / max,
    which is warp()'s clip=True)."""
    import numpy as np

    x = _frames(frames)
    if x.dtype not in (torch.float32, torch.float64):
        raise TypeError("warp_affine needs float32 or float64 frames")
    n, h, w = x.shape
    if order == 0:
        lo, hi = 0.0, 0.0
    elif lo is None or hi is None:
        lo, hi = minmax(x)
    lo = _per_frame(lo, n, x.device)[0].expand(n).contiguous()
    hi = _per_frame(hi, n, x.device)[0].expand(n).contiguous()
    m = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64)[:2, :3])
    out = torch.empty_like(x)
    check(_lib.load().pl_warp_affine(x.data_ptr(), out.data_ptr(), _dt(x), n, h, w, int(order), m.ctypes.data, lo.data_ptr(),
                                     hi.data_ptr(), _stream()), "pl_warp_affine")
    return out
