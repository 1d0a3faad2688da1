"""Drop-in mirror of ``pylinac.core.array_utils`` for the hot-path functions
(pylinac/core/array_utils.py:63-168): same names, argument meaning and error behaviour, with the
arithmetic done by libpylinac_hip.so on the GPU.

Inputs may be numpy arrays (1-D profiles or 2-D frames; the result is a numpy array, as in the
reference) or ``torch`` tensors already on the GPU (``[H,W]`` / ``[N,H,W]``; the result stays on
the device -- the batched fast path).  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

_NP_OK = (np.uint8, np.uint16, np.int16, np.int32, np.int64, np.float32, np.float64)
_TORCH_OF = {np.uint8: torch.uint8, np.uint16: torch.uint16, np.int16: torch.int16, np.int32: torch.int32,
             np.int64: torch.int64, np.float32: torch.float32, np.float64: torch.float64}


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "pylinac_amd needs a HIP device (torch.cuda.is_available() is False); "
            "there is no CPU fallback by design"
        )
    return torch.device("cuda", torch.cuda.current_device())


def array_not_empty(array) -> None:
    """pylinac/core/array_utils.py:23-26."""
    size = array.numel() if isinstance(array, torch.Tensor) else np.asarray(array).size
    if not size:
        raise ValueError("Array must not be empty")


_NP_OF_TORCH = {torch.uint8: np.dtype(np.uint8), torch.uint16: np.dtype(np.uint16), torch.int16: np.dtype(np.int16),
                torch.int32: np.dtype(np.int32), torch.int64: np.dtype(np.int64), torch.float32: np.dtype(np.float32),
                torch.float64: np.dtype(np.float64)}


class _Staged:
    """numpy <-> device staging: 1-D -> [1,1,L], 2-D -> [1,H,W]; tensors pass through."""

    def __init__(self, array):
        array_not_empty(array)
        self.is_tensor = isinstance(array, torch.Tensor)
        if self.is_tensor:
            self.ndim = array.dim()
            self.t = array
            return
        a = np.asarray(array)
        if a.dtype.type not in _NP_OK:
            if a.dtype == np.bool_:
                a = a.astype(np.uint8)
            elif a.dtype.kind in "iu":
                a = a.astype(np.int64)
            else:
                raise TypeError(f"unsupported dtype {a.dtype}")
        if a.ndim not in (1, 2):
            raise ValueError(f"expected a 1-D profile or 2-D frame; got {a.ndim}-D")
        self.ndim = a.ndim
        self.shape = a.shape
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a).to(_device())
        self.t = t.reshape(1, 1, -1) if a.ndim == 1 else t.unsqueeze(0)

    def out(self, t: torch.Tensor):
        if self.is_tensor:
            return t.reshape(self.t.shape) if t.numel() == self.t.numel() else t
        return t.cpu().numpy().reshape(self.shape)


def resolve_filter_size(length: int, size):
    """pylinac/core/array_utils.py:124-129: float in (0,1) -> int(round(len*size)) >= 1."""
    if isinstance(size, float):
        if 0 < size < 1:
            size = int(round(length * size))
            size = max(size, 1)
        else:
            raise ValueError("Float was passed but was not between 0 and 1")
    return size


def filter(array, size=0.05, kind: str = "median"):
    """Mirror of ``pylinac.core.array_utils.filter`` (array_utils.py:105-138).
    ``len(array)`` (the ROW count of a 2-D frame) scales a float ``size``."""
    s = _Staged(array)
    if s.is_tensor:
        length = s.t.shape[-2] if s.t.dim() >= 2 else s.t.shape[0]
    else:
        length = s.shape[0]
    size = resolve_filter_size(length, size)
    if kind == "median":
        return s.out(ops.median_filter(s.t, int(size)))
    elif kind == "gaussian":
        t = s.t
        if t.dim() == 3 and t.shape[1] == 1:  # 1-D profile: a single pass along the data axis
            return s.out(ops.gaussian_filter1d(t, size, axis=1))
        return s.out(ops.gaussian_filter(t, size))
    raise ValueError(f"Filter type {kind} unsupported. Use one of 'median', 'gaussian'")


def normalize(array, value=None):
    """array_utils.py:63-71."""
    s = _Staged(array)
    return s.out(ops.normalize(s.t, value))


def _refuse_inexact_64bit(array, what: str) -> None:
    """min / max travel as float64 scalars: 64-bit integers beyond 2^53 would lose their low bits there."""
    if not isinstance(array, torch.Tensor):
        a = np.asarray(array)
        if a.dtype in (np.int64, np.uint64) and a.size and (int(a.max()) > 2**53 or int(a.min()) < -(2**53)):
            raise TypeError(f"{what}: 64-bit integer values beyond 2**53 are not supported exactly on this backend")


def invert(array):
    """array_utils.py:74-77."""
    _refuse_inexact_64bit(array, "invert")
    s = _Staged(array)
    return s.out(ops.invert(s.t))


def ground(array, value: float = 0):
    """array_utils.py:92-102: ``array - array.min() + value``.  numpy's promotion of ``int_array + value``: a Python
    ``float`` (any, 0.0 and 1.0 included -- also inf / nan) or a numpy floating scalar turns an integer array into a float
    array (float64 for Python floats and float64 scalars), a Python ``int`` keeps the array's dtype."""
    _refuse_inexact_64bit(array, "ground")
    s = _Staged(array)
    t = s.t
    if not t.dtype.is_floating_point and isinstance(value, (float, np.floating)):
        if isinstance(value, np.floating) and value.dtype.itemsize < 8:
            # numpy subtracts in the INTEGER dtype (wrapping like `array - array.min()` does) and adds the narrow scalar in
            # the promoted narrow float type: a float64 detour would round twice and never wrap.  The device grounds in the
            # integer dtype (exact, wrapping); the one narrow add is numpy's own on the host.
            grounded = s.out(ops.ground(t, 0))
            if isinstance(grounded, torch.Tensor):
                raise TypeError("ground(): a numpy floating scalar narrower than float64 is supported for numpy input only")
            return grounded + value
        # numpy grounds in the INTEGER dtype first (`array - array.min()`, wrapping where the range exceeds the dtype: int16
        # frames spanning more than 32767 levels), THEN promotes for the float addition
        g = ops.normalize(ops.ground(t, 0), 1.0)                         # exact conversion of the grounded integers to float64
        return s.out(ops.ground(g, float(value), mn=torch.zeros(g.shape[0] if g.dim() == 3 else 1, dtype=torch.float64,
                                                                device=g.device)))
    return s.out(ops.ground(t, value))


def stretch(array, min: int = 0, max: int = 1):
    """array_utils.py:141-168: ``ground(normalize(ground(a)) * (max-min), value=min)``."""
    if max <= min:
        raise ValueError(f"Max must be larger than min. Passed max of {max} was <= {min}")
    a = array if isinstance(array, torch.Tensor) else np.asarray(array)
    info_dtype = _NP_OF_TORCH[a.dtype] if isinstance(a, torch.Tensor) else a.dtype     # no device-to-host copy for a dtype
    info = np.iinfo(info_dtype) if info_dtype.kind in "iu" else np.finfo(info_dtype)
    if max > info.max:
        raise ValueError(f"Max of {max} was larger than the allowed datatype maximum of {info.max}")
    if min < info.min:
        raise ValueError(f"Min of {min} was smaller than the allowed datatype minimum of {info.min}")
    s = _Staged(array)
    g = ops.ground(s.t)
    nrm = ops.normalize(g)
    scaled = ops.scale(nrm, float(max - min))
    return s.out(ops.ground(scaled, value=float(min)))


def bit_invert(array):
    """array_utils.py:80-89: ``np.invert`` = the datatype-specific complement (0 -> 255 for uint8, -1 for int8).  The
    complement is taken bitwise on the device (``pl_bit_invert``); 64-bit types keep every bit (uint64 travels as the
    int64 with the same bits, narrower unstaged types as int64 values whose low bits are the answer)."""
    a = np.asarray(array)
    array_not_empty(a)
    if a.dtype.kind not in "iub":
        raise ValueError(f"The datatype {a.dtype} could not be safely inverted. This usually means the array is a "
                         "float-like datatype. Cast to an integer-like datatype first.")
    if a.dtype == np.bool_:
        return np.logical_not(a)                     # np.invert on bool is logical not: no arithmetic to offload
    from ._lib import check, load

    src = a.view(np.int64) if a.dtype == np.uint64 else a
    s = _Staged(src)
    out = torch.empty_like(s.t)
    check(load().pl_bit_invert(s.t.data_ptr(), out.data_ptr(), ops._dt(s.t), s.t.numel(), ops._stream()), "pl_bit_invert")
    res = s.out(out)
    return res.view(np.uint64) if a.dtype == np.uint64 else res.astype(a.dtype)


def convert_to_dtype(array, dtype):
    """array_utils.py:171-198: rescale the VALUES to the same relative position in the new datatype's range (integer
    input: ``a / old_max``; float input: stretched to 0..1), then ``relative * range - max - 1`` cast to ``dtype``
    (the cast wraps negative values into unsigned types, which is what makes the formula land on ``relative * max``)."""
    a = np.asarray(array)
    array_not_empty(a)
    new = np.dtype(dtype)
    if a.dtype.kind == "f":
        rel = _Staged(stretch(a, min=0, max=1)).t
    else:
        rel = ops.normalize(_Staged(a).t, float(np.iinfo(a.dtype).max))
    info = np.iinfo(new) if new.kind in "iu" else np.finfo(new)
    f = ops.scale(rel, float(info.max) - float(info.min))
    zeros = torch.zeros(f.shape[0], dtype=torch.float64, device=f.device)
    f = ops.ground(ops.ground(f, value=-float(info.max), mn=zeros), value=-1.0, mn=zeros)     # (x - max) - 1
    kernel_dtype = new if new.type in _NP_OK else (np.dtype(np.int64) if new.kind in "iu" else np.dtype(np.float64))
    out = torch.empty(f.shape, dtype=_TORCH_OF[kernel_dtype.type], device=f.device)
    from ._lib import check, load

    check(load().pl_cast_wrap(f.data_ptr(), out.data_ptr(), ops._dt(out), f.numel(), ops._stream()), "pl_cast_wrap")
    return out.cpu().numpy().reshape(a.shape).astype(new)      # int8 / uint32 ...: exact narrowing of the int64 result


def geometric_center_idx(array) -> float:
    """array_utils.py:37-44 (pure index arithmetic)."""
    a = np.asarray(array)
    array_not_empty(a)
    if a.ndim > 1:
        raise ValueError(f"Array was multidimensional. Must pass 1D array; found {a.ndim}")
    return (a.shape[0] - 1) / 2.0


def geometric_center_value(array) -> float:
    """array_utils.py:47-60."""
    a = np.asarray(array)
    array_not_empty(a)
    if a.ndim > 1:
        raise ValueError(f"Array was multidimensional. Must pass 1D array; found {a.ndim}")
    n = a.shape[0]
    if n % 2 == 0:
        return (a[int(n / 2)] + a[int(n / 2) - 1]) / 2.0
    return a[int((n - 1) / 2)]
