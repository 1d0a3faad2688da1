"""TEST INFRASTRUCTURE ONLY: run pylinac_amd's host layer against the emulated kernel library (tests/hipemu).

`emulated_device()` is a context manager for the `-m "not gpu"` suite.  Inside it
  * pylinac_amd._lib hands out tests/hipemu/_build/libpylinac_emu.so (the csrc kernels compiled for the CPU fiber
    emulator) instead of libpylinac_hip.so,
  * CPU tensors answer `is_cuda == True`, so the host code takes its normal path and passes HOST pointers across the
    C ABI -- which is what the emulated kernels dereference,
  * torch.cuda.current_stream() returns a null stream handle.
The product itself contains no such switch: outside this context it refuses CPU tensors and fails without the HIP
library (tests/test_cabi.py).  Nothing here is evidence of GPU parity -- that is what the `-m gpu` tests are for; this
only lets host logic + kernel logic be exercised together where there is no GPU.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import sys
from pathlib import Path
from unittest import mock

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "hipemu"))


def _to_cpu(obj):
    if isinstance(obj, torch.device) and obj.type == "cuda":
        return torch.device("cpu")
    if isinstance(obj, str) and obj.startswith("cuda"):
        return "cpu"
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(o) for o in obj)
    return obj


class _CudaIsCpu(torch.overrides.TorchFunctionMode):
    """Every torch call that names a cuda device gets the cpu instead (factories, .to(), .cuda())."""

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = {k: _to_cpu(v) for k, v in (kwargs or {}).items()}
        if getattr(func, "__name__", "") == "cuda":       # Tensor.cuda()
            return args[0]
        return func(*_to_cpu(tuple(args)), **kwargs)


class _NullStream:
    cuda_stream = 0

    def synchronize(self):
        pass


def load_emulated_library():
    import build as emu_build  # tests/hipemu/build.py
    from pylinac_amd import _lib as binding

    lib = C.CDLL(str(emu_build.build()))
    for name, (argtypes, restype) in binding.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:       # a kernel file the emulator build leaves out: calling it must fail loudly
            continue
        fn.argtypes = argtypes
        fn.restype = restype
    lib.pl_last_error.restype = C.c_char_p
    lib.pl_status_string.restype = C.c_char_p
    return lib


@contextlib.contextmanager
def emulated_device():
    from pylinac_amd import _lib as binding

    lib = load_emulated_library()
    with contextlib.ExitStack() as stack:
        stack.enter_context(mock.patch.object(binding, "_lib", lib))
        stack.enter_context(mock.patch.object(torch.Tensor, "is_cuda", property(lambda self: True), create=True))
        stack.enter_context(mock.patch.object(torch.cuda, "current_stream", lambda device=None: _NullStream()))
        stack.enter_context(mock.patch.object(torch.cuda, "current_device", lambda: 0))
        stack.enter_context(mock.patch.object(torch.cuda, "synchronize", lambda device=None: None))
        stack.enter_context(mock.patch.object(torch.cuda, "is_available", lambda: True))
        stack.enter_context(_CudaIsCpu())
        yield lib
