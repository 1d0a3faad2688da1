"""CPU: the C-ABI library builds, loads, and exports exactly what include/pylinac_hip.h declares.
No compute call is made without a GPU (argument validation happens before any launch)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pylinac_hip.h")


@pytest.fixture(scope="module")
def lib():
    from pylinac_amd import _build, _lib

    _build.build()
    return _lib.load()


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pl_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds(lib):
    from pylinac_amd import _lib

    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(_lib.SIGNATURES) == syms
    for s in syms:
        assert hasattr(lib, s), s


def test_library_exports_match_header():
    from pylinac_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.lib_path())], capture_output=True, text=True,
                         check=True).stdout
    exported = sorted(set(re.findall(r" T (pl_[a-z0-9_]+)", out)))
    assert exported == header_symbols()


def test_library_contains_gfx950_code_object():
    from pylinac_amd import _lib

    data = open(_lib.lib_path(), "rb").read()
    assert b"gfx950" in data


def test_status_strings_and_version(lib):
    from pylinac_amd import _lib

    assert lib.pl_abi_version() == _lib.ABI_VERSION == 3
    assert lib.pl_status_string(0) == b"ok"
    assert b"invalid" in lib.pl_status_string(1)


def test_argument_validation_without_launch(lib):
    """Invalid arguments are rejected with PL_ERR_INVALID_ARG before any HIP call."""
    assert lib.pl_gaussian1d(None, None, 0, 1, 4, 4, 0, None, None, 1, None) == 1
    assert b"pl_gaussian1d" in lib.pl_last_error()
    buf = (C.c_uint16 * 16)()
    w = (C.c_double * 3)()
    p = C.cast(buf, C.c_void_p)
    assert lib.pl_gaussian1d(p, p, 0, 1, 4, 4, 0, C.cast(w, C.c_void_p), None, 1, None) == 1  # in-place refused
    assert lib.pl_median2d(p, p, 0, 1, 4, 4, 3, None) == 1
    assert lib.pl_hist16(p, 3, 1, 16, p, None) == 1  # float dtype refused
    assert lib.pl_reduce_axis(p, 0, 1, 4, 4, 2, 0, p, None) == 1  # bad axis
    assert lib.pl_threshold(p, p, 0, 1, 16, None, 0, 0, None) == 1
    # pl_hist16_wl: ranks and their output go together; the scratch table of the order-statistics form is read back in quads
    big = (C.c_uint32 * 32)()
    base = C.addressof(big)
    aligned, odd = C.c_void_p((base + 15) & ~15), C.c_void_p(((base + 15) & ~15) + 4)
    assert lib.pl_hist16_wl(p, 0, 1, 4, 4, aligned, p, 2, p, p, p, 0, None, None) == 1 and b"go together" in lib.pl_last_error()
    assert lib.pl_hist16_wl(p, 0, 1, 4, 4, odd, p, 2, p, p, p, 1, p, None) == 1 and b"16-byte" in lib.pl_last_error()
    assert lib.pl_hist16_wl(p, 0, 1, 4, 4, aligned, p, 0, p, p, None, 0, None, None) == 1   # an edge window of zero pixels


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pylinac_amd import _lib

    monkeypatch.setenv("PYLINAC_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.PylinacHipError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under pylinac_amd/ may import it."""
    pkg = os.path.join(ROOT, "pylinac_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                # nor the CPU kernel emulator of tests/hipemu (test infrastructure as well)
                assert not re.search(r"hipemu|emu_backend|PL_EMULATE", text), f
                # scipy may only supply the reference's own per-profile / per-dataset optimisers (SURVEY section 8 marks
                # them host-side): the Hill and "top" fits of SingleProfile and Starshot's Nelder-Mead wobble circle.
                # No scipy.ndimage / scipy.signal / scipy.interpolate: those are what the kernels replace.
                for m in re.finditer(r"^\s*(?:from\s+(scipy[\w.]*)\s+import\s+([\w, ]+)|import\s+(scipy[\w.]*))", text, flags=re.M):
                    what = (m.group(1) or m.group(3), (m.group(2) or "").replace(" ", ""))
                    assert what in (("scipy.optimize", "curve_fit"), ("scipy.optimize", "minimize"), ("scipy", "optimize")), (f, what)


def test_no_cpu_fallback_without_device():
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pylinac_amd import array_utils as au

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        au.filter(np.arange(10), 3)
