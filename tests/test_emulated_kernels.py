"""Kernel LOGIC on the CPU: csrc/*.hip compiled for the wave64 fiber emulator of tests/hipemu and driven through the
same C ABI with host pointers, against the oracle.  This is NOT a product path (the product has no CPU path; see
tests/test_cabi.py) -- it lets kernel changes be checked where there is no GPU.  The `-m gpu` parity tests remain the
proof on hardware.  The first tests re-run kernels that are parity-green on the MI355X: they validate the emulator.
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "hipemu"))

from oracle import pylinac_oracle as orc  # noqa: E402
from pylinac_amd import _lib as binding  # noqa: E402  (argtypes only; the emulated library is loaded here, not there)

PL_U16, PL_I16, PL_F32, PL_F64, PL_U8, PL_I32, PL_I64 = 0, 1, 2, 3, 4, 5, 6
_DT = {np.dtype(np.uint16): PL_U16, np.dtype(np.int16): PL_I16, np.dtype(np.float32): PL_F32,
       np.dtype(np.float64): PL_F64, np.dtype(np.uint8): PL_U8, np.dtype(np.int32): PL_I32, np.dtype(np.int64): PL_I64}


@pytest.fixture(scope="module")
def emu():
    import build as emu_build  # tests/hipemu/build.py

    lib = C.CDLL(str(emu_build.build()))
    for name, (argtypes, restype) in binding.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = restype
    lib.pl_last_error.restype = C.c_char_p
    return lib


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ok(lib, rc):
    assert rc == 0, lib.pl_last_error().decode()


# ---- emulator validation on kernels that are parity-green on the GPU ---------------------------------------

def test_emu_interp1d_linear_and_cubic(emu):
    from scipy.interpolate import interp1d

    rng = np.random.default_rng(0)
    x = np.cumsum(rng.uniform(0.5, 1.5, 40))
    y = rng.normal(size=(3, 40))
    xq = np.linspace(x[0] - 1.0, x[-1] + 1.0, 333)
    for kind, name in ((0, "linear"), (1, "cubic")):
        out = np.empty((3, xq.size))
        work = np.empty(3 * 3 * 40)
        _ok(emu, emu.pl_interp1d(_p(x), 0, _p(y), 3, 40, _p(xq), xq.size, kind, _p(work), _p(out), None))
        ref = np.stack([interp1d(x, yy, kind=name, bounds_error=False, fill_value="extrapolate")(xq) for yy in y])
        if kind == 0:
            np.testing.assert_array_equal(out, ref)
        else:
            np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)


def test_emu_roi_stats_barriers_ballot_dynamic_lds(emu):
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 4000, (2, 64, 80)).astype(np.uint16)
    rois = np.array([[30.2, 31.7, 9.3, 0], [50.0, 20.0, 5.0, 0], [12.5, 40.5, 11.0, 0]])
    out = np.empty((2, 3, 6))
    status = np.empty((2, 3), np.int32)
    _ok(emu, emu.pl_roi_stats(_p(frames), PL_U16, 2, 64, 80, _p(rois), 3, 0, 0, _p(out), _p(status), None))
    assert (status == 0).all()
    for f in range(2):
        for k, (cx, cy, r, _) in enumerate(rois):
            want = orc.disk_roi_stats(frames[f], cx, cy, r)
            np.testing.assert_array_equal(out[f, k][[0, 3, 4, 5]], want[[0, 3, 4, 5]])
            np.testing.assert_allclose(out[f, k][[1, 2]], want[[1, 2]], rtol=1e-12)   # summation order


def test_emu_gamma2d(emu):
    rng = np.random.default_rng(2)
    ref = rng.uniform(0, 100, (1, 24, 28))
    ev = ref * rng.uniform(0.97, 1.03, ref.shape)
    dta = 2
    want = orc.gamma_2d(ref[0], ev[0], dose_to_agreement=2, distance_to_agreement=dta, gamma_cap_value=2,
                        global_dose=True, dose_threshold=5, fill_value=np.nan)
    from pylinac_amd.gamma import disk_offsets

    rr, cc = disk_offsets(dta)
    dr, dc = rr.astype(np.int32), cc.astype(np.int32)
    d2 = np.ascontiguousarray((rr / dta) ** 2 + (cc / dta) ** 2, dtype=np.float64)
    work = np.empty(2 * ref.size)
    out = np.empty_like(ref)
    rmax = np.array([ref.max()])
    _ok(emu, emu.pl_gamma2d(_p(ref), _p(ev), 1, 24, 28, 0.02, 1, _p(rmax), _p(dr), _p(dc), _p(d2), dr.size,
                            0.05, 2.0, float("nan"), _p(work), _p(out), None))
    np.testing.assert_array_equal(out[0], want)


def test_emu_xim_decode_scans(emu):
    rng = np.random.default_rng(3)
    h, w = 37, 53
    img = np.cumsum(rng.integers(-300, 300, (h, w)), axis=1).astype(np.int32)
    lookup, stream = orc.xim_encode(img)
    want = orc.xim_decode(lookup, stream, w, h, 4)
    np.testing.assert_array_equal(want.reshape(h, w), img)
    work = np.empty(emu.pl_xim_work_bytes(w, h), np.uint8)
    out = np.empty((h, w), np.int32)
    _ok(emu, emu.pl_xim_decode(_p(lookup), lookup.size, _p(stream), stream.size, w, h, 4, _p(out), _p(work), None))
    np.testing.assert_array_equal(out, img)


def test_emu_gaussian_f64_kernels_match_scipy(emu):
    from scipy import ndimage

    rng = np.random.default_rng(4)
    frames = rng.integers(0, 65535, (2, 70, 96)).astype(np.uint16)
    for sigma in (1.0, 2.3):
        radius = int(4.0 * sigma + 0.5)
        x = np.arange(-radius, radius + 1)
        wts = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        wts = np.ascontiguousarray(wts / wts.sum())
        out = np.empty_like(frames)
        tmp = np.empty_like(frames)
        _ok(emu, emu.pl_gaussian2d(_p(frames), _p(out), _p(tmp), PL_U16, 2, 70, 96, _p(wts), _p(wts), radius, None))
        want = np.stack([ndimage.gaussian_filter(f, sigma) for f in frames])
        np.testing.assert_array_equal(out, want)


def test_emu_gaussian_packed_f32_decision_kernels(emu):
    """gaussian_rw.hip (the bench's default Gaussian for 16-bit frames) under the emulator (needs the ROCm clang++ as
    host compiler: clang vector types): radii 4 / 8 / 12 / 20, uint16 and int16, frames with flat and saturated
    regions so that the LDS fix list and the exact float64 tiers run too.  Bit-identical to scipy."""
    import build as emu_build
    from scipy import ndimage

    if "gaussian_rw.hip" not in emu_build.SOURCES:
        pytest.skip("no clang++ host compiler: gaussian_rw.hip is not in the emulated library")
    rng = np.random.default_rng(8)
    base = rng.integers(0, 65535, (2, 96, 128)).astype(np.uint16)
    base[0, 20:60, 30:100] = 40000          # flat: S = c * sum(w) = c +- 1e-12 -> the undecided tiers
    base[1, :48, :] = 65535                 # saturated
    base[1, 60:, 64:] = 0
    smooth = (np.clip(ndimage.gaussian_filter(base.astype(float), 6), 0, 65535)).astype(np.uint16)
    for frames in (base, smooth, (smooth.astype(np.int32) - 32768).astype(np.int16)):
        for sigma in (1, 2, 3, 5):
            radius = int(4.0 * sigma + 0.5)
            x = np.arange(-radius, radius + 1)
            wts = np.exp(-0.5 / (sigma * sigma) * x ** 2)
            wts = np.ascontiguousarray(wts / wts.sum())
            out, tmp = np.empty_like(frames), np.empty_like(frames)
            _ok(emu, emu.pl_gaussian2d(_p(frames), _p(out), _p(tmp), _DT[frames.dtype], 2, 96, 128, _p(wts), _p(wts), radius, None))
            want = np.stack([ndimage.gaussian_filter(f, sigma) for f in frames])
            np.testing.assert_array_equal(out, want, err_msg=f"{frames.dtype} sigma {sigma}")


def test_emu_gaussian_register_window_kernels(emu):
    """gaussian_rw.hip (register-window kernels: lane-local minimum, wave-owned windows, raw LDS copy for the
    fix-ups): partial row tiles, two column tiles with a partial second one on axis 1 (right halo inside inactive lanes'
    positions), an odd number of rows, frames shorter than the halo (multiple reflections), zero / flat / saturated regions
    (list overflow -> whole-tile recompute), int16.  Bit-identical to scipy."""
    import build as emu_build
    from scipy import ndimage

    if "gaussian_rw.hip" not in emu_build.SOURCES:
        pytest.skip("no clang++ host compiler: gaussian_rw.hip is not in the emulated library")
    rng = np.random.default_rng(18)
    cases = []
    for shape in ((2, 150, 144), (1, 37, 1040), (3, 33, 16), (1, 9, 32)):
        smooth = ndimage.gaussian_filter(rng.integers(0, 65535, shape).astype(float), (0, 5, 5))
        a = np.clip(smooth + rng.normal(0, 300, shape), 0, 65535).astype(np.uint16)
        a[0, : shape[1] // 3, : shape[2] // 2] = 0
        a[-1, shape[1] // 2:, shape[2] // 2:] = 41234
        cases.append(a)
    cases.append((cases[0].astype(np.int32) - 31000).astype(np.int16))
    cases.append(np.full((1, 70, 128), 65535, dtype=np.uint16))
    for frames in cases:
        n, h, w = frames.shape
        for sigma in ((5,) if w > 1000 else (1, 2, 3, 5)):
            radius = int(4.0 * sigma + 0.5)
            x = np.arange(-radius, radius + 1)
            wts = np.exp(-0.5 / (sigma * sigma) * x ** 2)
            wts = np.ascontiguousarray(wts / wts.sum())
            out, tmp = np.empty_like(frames), np.empty_like(frames)
            _ok(emu, emu.pl_gaussian2d(_p(frames), _p(out), _p(tmp), _DT[frames.dtype], n, h, w, _p(wts), _p(wts), radius, None))
            want = np.stack([ndimage.gaussian_filter(f, sigma) for f in frames])
            np.testing.assert_array_equal(out, want, err_msg=f"{frames.shape} {frames.dtype} sigma {sigma}")


def test_emu_gaussian_marching_strip_kernel(emu):
    """gaussian_mm.hip, gauss2d_mm (both axes in one launch: marching strip, integer MFMA digit planes, carry-cascade
    decision, pair exchange for the stores): two strips with a partial second one (also ragged widths, 66 / 1046 with a quad
    across the right edge, 300), two row segments, mirrored border quads on both sides, zero / constant / saturated blocks (whole-tile constant path and per-output recompute), int16, all
    sigmas.  Bit-identical to scipy.  (Frames below 64 x 64 take the single-axis kernels: covered by the tests above.)"""
    import build as emu_build
    from scipy import ndimage

    if "gaussian_mm.hip" not in emu_build.SOURCES:
        pytest.skip("no clang++ host compiler: gaussian_mm.hip is not in the emulated library")
    rng = np.random.default_rng(31)
    cases = []
    for shape in ((2, 150, 144), (1, 300, 272), (1, 70, 1040), (2, 70, 66), (1, 80, 1046), (1, 90, 300)):
        smooth = ndimage.gaussian_filter(rng.integers(0, 65535, shape).astype(float), (0, 5, 5))
        a = np.clip(smooth + rng.normal(0, 300, shape), 0, 65535).astype(np.uint16)
        a[0, : shape[1] // 3, : shape[2] // 2] = 0
        a[-1, shape[1] // 2:, shape[2] // 2:] = 41234
        cases.append(a)
    cases.append((cases[0].astype(np.int32) - 31000).astype(np.int16))
    cases.append(np.full((1, 70, 128), 65535, dtype=np.uint16))
    for frames in cases:
        n, h, w = frames.shape
        for sigma in ((5,) if w > 1000 else (1, 2, 3, 5)):
            radius = int(4.0 * sigma + 0.5)
            x = np.arange(-radius, radius + 1)
            wts = np.exp(-0.5 / (sigma * sigma) * x ** 2)
            wts = np.ascontiguousarray(wts / wts.sum())
            out, tmp = np.full_like(frames, 12345), np.empty_like(frames)
            _ok(emu, emu.pl_gaussian2d(_p(frames), _p(out), _p(tmp), _DT[frames.dtype], n, h, w, _p(wts), _p(wts), radius, None))
            want = np.stack([ndimage.gaussian_filter(f, sigma) for f in frames])
            np.testing.assert_array_equal(out, want, err_msg=f"{frames.shape} {frames.dtype} sigma {sigma}")


def test_emu_gaussian_matrix_core_extreme_values(emu):
    """gauss2d_mm where the digit planes and the accumulators are at their extremes: full-range int16 noise, frames of only
    -32768 / 32767 and of only 0 / 65535 (every digit +-128 / 127), a saturated block on a saturated background; sigmas on
    both sides of the tap limit (0.9 .. 6: radius 4 .. 24, non-integer sigma included).  Bit-identical to scipy."""
    import build as emu_build
    from scipy import ndimage

    if "gaussian_mm.hip" not in emu_build.SOURCES:
        pytest.skip("no clang++ host compiler: gaussian_mm.hip is not in the emulated library")
    rng = np.random.default_rng(3)
    block = np.full((1, 64, 64), -32768, np.int16)
    block[0, 20:40, 10:50] = 32767
    cases = [rng.integers(-32768, 32768, (2, 96, 144)).astype(np.int16),
             np.where(rng.random((2, 80, 128)) < 0.5, -32768, 32767).astype(np.int16),
             np.where(rng.random((1, 70, 160)) < 0.5, 0, 65535).astype(np.uint16), block]
    for frames in cases:
        n, h, w = frames.shape
        for sigma in (0.9, 2, 3.3, 6):
            radius = int(4.0 * sigma + 0.5)
            x = np.arange(-radius, radius + 1)
            wts = np.exp(-0.5 / (sigma * sigma) * x ** 2)
            wts = np.ascontiguousarray(wts / wts.sum())
            out, tmp = np.empty_like(frames), np.empty_like(frames)
            _ok(emu, emu.pl_gaussian2d(_p(frames), _p(out), _p(tmp), _DT[frames.dtype], n, h, w, _p(wts), _p(wts), radius, None))
            want = np.stack([ndimage.gaussian_filter(f, sigma) for f in frames])
            np.testing.assert_array_equal(out, want, err_msg=f"{frames.shape} {frames.dtype} sigma {sigma}")


def test_emu_median_consumed_on_the_fly(emu):
    """pl_median3_otsu16 / pl_median3_threshold_colsum_u16 (median3_rows.h inside the Otsu window kernel and inside the
    threshold + column-sum kernel: the median plane is never written): tiny and ragged geometries (2 rows, one 8-column
    block, widths that leave the last wave partly idle, heights off the 16 / 32 row groups), a full-range frame that does not
    fit the one-pass window (flagged: the full-range kernel with packed 16-bit counters takes it), int16."""
    from scipy import ndimage

    rng = np.random.default_rng(5)
    # (2, 70, 1040): 72 800 pixels per frame -- a batch this small gets EIGHT workgroups per frame, each tallying a band of
    # rows into its own LDS window, merged through the frame's table; the second frame spills and goes to the full-range kernel
    for shape in ((1, 2, 8), (2, 3, 16), (1, 5, 520), (2, 33, 24), (1, 40, 1032), (3, 17, 64), (2, 70, 1040)):
        for dt, code in ((np.uint16, PL_U16), (np.int16, PL_I16)):
            n, h, w = shape
            a = (rng.integers(1000, 1400, shape) + (np.arange(w) > w // 2) * 5000).astype(np.int64)
            a[-1] = rng.integers(0, 65536, (h, w))
            a = (a - (32768 if dt == np.int16 else 0)).astype(dt)
            med = np.stack([ndimage.median_filter(f, size=3) for f in a])
            thr, mn, mx, flag = (np.zeros(n, np.int32) for _ in range(4))
            hist, scratch = np.zeros((n, 65536), np.uint32), np.zeros_like(a)
            _ok(emu, emu.pl_median3_otsu16(_p(a), _p(scratch), code, n, h, w, None, None, _p(thr), _p(mn), _p(mx), _p(flag),
                                           _p(hist), None))
            np.testing.assert_array_equal(thr, [orc.threshold_otsu(f) for f in med], err_msg=f"{shape} {dt.__name__}")
            np.testing.assert_array_equal(mn, med.reshape(n, -1).min(1))
            np.testing.assert_array_equal(mx, med.reshape(n, -1).max(1))
            if h * w >= 1000:
                assert flag[-1] == 1, (shape, flag)                        # the full-range frame went to the full-range kernel
            if dt == np.uint16:
                cut = np.array([int(np.percentile(f, 40)) for f in med], np.int32)
                out, cs = np.zeros_like(a), np.zeros((n, w), np.uint64)
                _ok(emu, emu.pl_median3_threshold_colsum_u16(_p(a), _p(out), n, h, w, _p(cut), _p(cs), None))
                want = np.where(med.astype(np.int64) >= cut[:, None, None], med, 0).astype(np.uint16)
                np.testing.assert_array_equal(out, want, err_msg=str(shape))
                np.testing.assert_array_equal(cs.astype(np.int64), want.astype(np.int64).sum(1))


def test_emu_full_range_otsu_counter_overflow(emu):
    """otsu16_full_kernel (the fallback of pl_otsu16 / pl_median3_otsu16 for frames wider than the 38 912-bin window): 65 536
    packed 16-bit counters whose guard bit folds 32 768 counts away at a time.  Frames of > 32 768 pixels with one value
    holding most of them, scattered (every add is a lane's own: the guard is crossed by single adds) and as flat areas (the
    wave-uniform bulk add crosses it), twice over (two folds of one key), next to full-range noise; plain and median paths."""
    from scipy import ndimage

    rng = np.random.default_rng(9)
    h, w = 96, 1032                                                     # 99 072 pixels
    for dt, code in ((np.uint16, PL_U16), (np.int16, PL_I16)):
        off = 32768 if dt == np.int16 else 0
        noise = rng.integers(0, 65536, (3, h, w))
        a = noise.copy()
        a[0][rng.random((h, w)) < 0.75] = 777                           # scattered: ~74 000 single adds to one bin (two folds)
        a[1][:, : w // 2] = 40000                                       # flat half: wave-uniform bulk adds
        a[1][rng.random((h, w)) < 0.3] = 40001                          # ... broken up by a second heavy value
        a[2] = np.where(rng.random((h, w)) < 0.5, 12, 65535)            # two values only, at the ends of the range
        a = (a - off).astype(dt)
        n = a.shape[0]
        thr, mn, mx, flag = (np.zeros(n, np.int32) for _ in range(4))
        hist = np.zeros((n, 65536), np.uint32)
        _ok(emu, emu.pl_otsu16(_p(a), code, n, h * w, None, None, _p(thr), _p(mn), _p(mx), _p(flag), _p(hist), None))
        assert flag.all()
        np.testing.assert_array_equal(thr, [orc.threshold_otsu(f) for f in a], err_msg=dt.__name__)
        np.testing.assert_array_equal(mn, a.reshape(n, -1).min(1))
        np.testing.assert_array_equal(mx, a.reshape(n, -1).max(1))
        # (the window kernel's several-workgroups-per-frame form on frames that DO fit the window, plain path: 2 frames of
        # 99 072 pixels get eight parts each, a ninth-of-a-vector tail included)
        b = (rng.integers(3000, 9000, (2, h * w + 5)) - off).astype(dt)
        t2, mn2, mx2, f2 = (np.zeros(2, np.int32) for _ in range(4))
        _ok(emu, emu.pl_otsu16(_p(b), code, 2, h * w + 5, None, None, _p(t2), _p(mn2), _p(mx2), _p(f2), _p(hist), None))
        assert not f2.any()
        np.testing.assert_array_equal(t2, [orc.threshold_otsu(f) for f in b])
        np.testing.assert_array_equal(mn2, b.min(1))
        np.testing.assert_array_equal(mx2, b.max(1))
        med = np.stack([ndimage.median_filter(f, size=3) for f in a])
        scratch = np.zeros_like(a)
        _ok(emu, emu.pl_median3_otsu16(_p(a), _p(scratch), code, n, h, w, None, None, _p(thr), _p(mn), _p(mx), _p(flag), _p(hist), None))
        assert flag.all()
        np.testing.assert_array_equal(thr, [orc.threshold_otsu(f) for f in med], err_msg=dt.__name__ + " median")
        np.testing.assert_array_equal(mn, med.reshape(n, -1).min(1))
        np.testing.assert_array_equal(mx, med.reshape(n, -1).max(1))


def test_emu_median3_packed_kernels(emu):
    from scipy import ndimage

    rng = np.random.default_rng(5)
    for shape in ((2, 40, 64), (1, 33, 50), (1, 21, 37)):     # eight-column, pair and generic paths
        frames = rng.integers(0, 65535, shape).astype(np.uint16)
        out = np.empty_like(frames)
        _ok(emu, emu.pl_median2d(_p(frames), _p(out), PL_U16, shape[0], shape[1], shape[2], 3, None))
        np.testing.assert_array_equal(out, np.stack([ndimage.median_filter(f, size=3) for f in frames]))


def _label_masks():
    """random masks of several densities and shapes (frame sizes that are / are not multiples of the 64-lane chunk),
    plus structured ones: solid, a frame with one hole, stripes, a checkerboard, staircases, a spiral"""
    rng = np.random.default_rng(6)
    out = []
    for shape in ((2, 45, 61), (3, 16, 64), (2, 33, 130), (1, 70, 7), (2, 1, 200), (2, 9, 1)):
        for dens in (0.15, 0.45, 0.6, 0.9):
            out.append((rng.random(shape) < dens).astype(np.uint8))
    h, w = 40, 150
    solid = np.ones((1, h, w), np.uint8)
    ring = solid.copy(); ring[0, 10:30, 20:120] = 0; ring[0, 15:25, 40:100] = 1
    stripes_v = np.zeros((1, h, w), np.uint8); stripes_v[0, :, ::3] = 1
    stripes_h = np.zeros((1, h, w), np.uint8); stripes_h[0, ::2, :] = 1
    checker = (np.indices((h, w)).sum(0) % 2).astype(np.uint8)[None]
    stairs = np.zeros((1, h, w), np.uint8)
    for k in range(h):
        stairs[0, k, max(0, 3 * k - 4):3 * k + 2] = 1
    anti = stairs[:, :, ::-1].copy()
    spiral = np.zeros((1, 41, 41), np.uint8)
    r, c, dr, dc, n = 0, 0, 0, 1, 41
    seg = [41, 40, 40, 38, 38, 36, 36, 34, 34, 32, 32, 30, 30, 28, 28, 26, 26, 24, 24, 22, 22]
    for k, length in enumerate(seg):
        for _ in range(length - (0 if k == 0 else 1)):
            spiral[0, r, c] = 1
            r, c = r + dr, c + dc
        r, c = r - dr, c - dc
        dr, dc = dc, -dr
        r, c = r + dr, c + dc
    out += [solid, ring, stripes_v, stripes_h, checker, stairs, anti, spiral, 1 - spiral, 1 - ring]
    return out


def _check_label(emu):
    from scipy import ndimage

    for mask in _label_masks():
        n, h, w = mask.shape
        for conn, structure in ((4, None), (8, np.ones((3, 3)))):
            labels = np.empty(mask.shape, np.int32)
            work = np.empty(mask.shape, np.int32)
            count = np.empty(n, np.int32)
            _ok(emu, emu.pl_label(_p(mask), n, h, w, conn, _p(labels), _p(work), _p(count), None))
            for f in range(n):
                want, nl = ndimage.label(mask[f], structure=structure)
                assert count[f] == nl, (mask.shape, conn)
                np.testing.assert_array_equal(labels[f], want, err_msg=f"{mask.shape} conn {conn}")


def _check_fill_holes_and_clear_border(emu):
    from scipy import ndimage

    for mask in _label_masks():
        n, h, w = mask.shape
        work = np.empty(mask.shape, np.int32)
        flags = np.empty(mask.shape, np.uint8)
        for conn_bg, structure in ((4, None), (8, np.ones((3, 3)))):
            out = np.empty_like(mask)
            _ok(emu, emu.pl_fill_holes(_p(mask), _p(out), n, h, w, conn_bg, _p(work), _p(flags), None))
            for f in range(n):
                # scipy's `structure` is the connectivity of the hole-growing step, i.e. of the BACKGROUND
                want = ndimage.binary_fill_holes(mask[f], structure=structure)
                np.testing.assert_array_equal(out[f].astype(bool), want, err_msg=f"fill {mask.shape} {conn_bg}")
        for buf in (0, 2):
            out = np.empty_like(mask)
            _ok(emu, emu.pl_clear_border(_p(mask), _p(out), n, h, w, buf, _p(work), _p(flags), None))
            for f in range(n):
                want = orc.clear_border_like_skimage(mask[f].astype(bool), buf)
                np.testing.assert_array_equal(out[f].astype(bool), want, err_msg=f"clear {mask.shape} {buf}")


# ---- host layer + kernels together on the emulated device (tests/emu_backend.py) ---------------------------------
# Same check functions as the `-m gpu` tests (tests/next_row_checks.py); sized for the fiber emulator.

@pytest.fixture()
def emulated():
    from emu_backend import emulated_device

    with emulated_device():
        import torch

        yield torch.device("cuda:0")


def test_emulated_hough_line_peaks(golden, emulated):
    import next_row_checks as checks

    checks.check_hough_line_peaks(golden, emulated)


def test_emulated_phantom_outline(golden, emulated):
    import next_row_checks as checks

    checks.check_phantom_outline(golden, emulated, names=["sq0"])


def test_emulated_region_moments(golden, emulated):
    import next_row_checks as checks

    checks.check_region_moments_kernel(golden, emulated)
    checks.check_phantom_outline(golden, emulated, names=["sq45"])      # the symmetric region of round 1's failure


def test_emulated_otsu16(golden, emulated):
    import next_row_checks as checks

    checks.check_otsu16(golden, emulated, big=False)


def test_emulated_ctp528_batch(golden, emulated):
    import next_row_checks as checks

    checks.check_ctp528_batch(golden, emulated, whole=False)


def test_emulated_field_cax(emulated):
    import next_row_checks as checks

    checks.check_field_cax(emulated)


def test_emulated_wl_analyze_batch(golden, emulated):
    import next_row_checks as checks

    checks.check_wl_analyze_batch(golden, emulated, frames=[0, 6, 7])      # plain, inverted, edge-cleaned


def test_emulated_rectangle_roi(golden, emulated):
    import next_row_checks as checks

    checks.check_rectangle_roi(golden, emulated)


def test_emulated_single_profile_hill_and_penumbra(golden, emulated):
    """SingleProfile (Hill edge, penumbra for the three edge methods) on the emulated device, a subset of hill.npz."""
    import next_row_checks as checks
    from pylinac_amd import profile

    def make(values, edge, **kw):
        return profile.SingleProfile(values, edge_detection_method=edge, **kw)

    n = checks.check_hill_and_penumbra(golden("hill"), make, tol=1e-9, spline_tol=1e-6,
                                       only=lambda t: t.startswith(("fx3.", "fx12.", "epid.", "fff0.", "fff2.")))
    assert n >= 20


def test_emulated_hill_fit_matches_scipy(emulated):
    """pl_hill_fit (MINPACK lmdif restated, one lane per fit) against scipy's leastsq on synthetic penumbra windows."""
    import next_row_checks as checks
    from pylinac_amd import ops

    import torch

    def fit(xs, ys, lens):
        dev = torch.device("cuda:0")
        p, info, nfev = ops.hill_fit(torch.from_numpy(xs).to(dev), torch.from_numpy(ys).to(dev), torch.from_numpy(lens).to(dev))
        return p.cpu().numpy(), info.cpu().numpy(), nfev.cpu().numpy()

    def fit_ex(xs, ys, lens):
        dev = torch.device("cuda:0")
        p, info, nfev, step = ops.hill_fit(torch.from_numpy(xs).to(dev), torch.from_numpy(ys).to(dev), torch.from_numpy(lens).to(dev),
                                           last_step=True)
        return p.cpu().numpy(), info.cpu().numpy(), nfev.cpu().numpy(), step.cpu().numpy()

    assert checks.check_hill_fit_vs_scipy(fit_ex, n=40) >= 36
    assert checks.check_hill_fit_kernels_agree(fit, n=24) >= 18
    assert checks.check_hill_fit_pathological(fit, fit_ex)


def test_emu_hill_entry_points_reject_bad_arguments(emu):
    """The Hill entry points' own argument checks (C ABI: PL_REQUIRE -> non-zero code + pl_last_error), no launch."""
    d = np.zeros(64)
    i32 = np.zeros(16, np.int32)
    f = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    emu.pl_hill_fit.argtypes, emu.pl_hill_fit.restype = f, C.c_int
    assert emu.pl_hill_fit(_p(d), _p(d), None, 1, 3, 3, _p(d), _p(d), _p(i32), None, None) != 0          # fewer than 4 samples of room
    assert b"4 .. 1024" in emu.pl_last_error()
    assert emu.pl_hill_fit(_p(d), _p(d), None, 1, 8, 4, _p(d), _p(d), _p(i32), None, None) != 0          # stride < mmax
    assert emu.pl_hill_fit(None, _p(d), None, 1, 8, 8, _p(d), _p(d), _p(i32), None, None) != 0           # null pointer
    assert emu.pl_hill_fit(_p(d), _p(d), None, 0, 8, 8, _p(d), _p(d), _p(i32), None, None) == 0          # empty batch: nothing to do
    emu.pl_hill_penumbra.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    emu.pl_hill_penumbra.restype = C.c_int
    assert emu.pl_hill_penumbra(_p(d), _p(d), 1, 80.0, 20.0, _p(d), None) != 0                           # lower > upper
    emu.pl_profile_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    emu.pl_profile_lookup.restype = C.c_int
    assert emu.pl_profile_lookup(_p(d), _p(d), 1, 1, _p(d), 1, _p(d), None) != 0                         # one sample: no interval


def test_emulated_hill_batch(golden, emulated):
    """single_profile_hill_batch (no per-profile host call) against the reference's SingleProfile numbers of hill.npz and
    against the per-profile mirror."""
    import next_row_checks as checks
    from pylinac_amd import profile

    n = checks.check_hill_batch(golden("hill"), profile.single_profile_hill_batch,
                                only=lambda t: t.startswith(("fx0.", "fx2.", "fx3.", "fx6.", "fx8.", "epid.hill.dpmm", "fff1.")))
    assert n >= 10
    single = lambda v, **kw: profile.SingleProfile(v, edge_detection_method=profile.Edge.INFLECTION_HILL, **kw)
    checks.check_hill_batch_vs_single(profile.single_profile_hill_batch, single, n=3, length=90)
    assert checks.check_hill_batch_options(profile.single_profile_hill_batch, single) == 36


def test_emulated_fwhm_batch(golden, emulated):
    """single_profile_fwhm_batch (the default edge method for every row of a batch) against the reference's own SingleProfile
    numbers (its frozen 63- and 65-detector profiles) and against the per-profile SingleProfile."""
    import next_row_checks as checks
    from pylinac_amd import profile

    assert checks.check_fwhm_batch_golden(golden("single_profile"), profile.single_profile_fwhm_batch, only_lengths={63, 65}) == 21
    fns = dict(infl=profile.single_profile_inflection_batch, fwhm=profile.single_profile_fwhm_batch,
               hill=profile.single_profile_hill_batch)
    assert checks.check_profile_batch_golden(golden("profile_batch"), fns) == 33

    assert checks.check_fwhm_batch(profile.single_profile_fwhm_batch, lambda v, **kw: profile.SingleProfile(v, **kw),
                                   xs=(50, 20), norms=("Geometric center", "Beam center")) == 36
    assert checks.check_inflection_batch(
        profile.single_profile_inflection_batch,
        lambda v, **kw: profile.SingleProfile(v, edge_detection_method=profile.Edge.INFLECTION_DERIVATIVE, **kw),
        norms=(None, "Beam center")) == 18


def test_emulated_starshot(golden, emulated):
    """Starshot.analyze on the emulated device (one integer and the float32 frame; the full set runs with -m gpu)."""
    import next_row_checks as checks
    from pylinac_amd import starshot

    n = checks.check_starshot(golden("starshot"), lambda f, dpi, sid: starshot.Starshot(f, dpi=dpi, sid=sid),
                              only=("inverted", "float"))
    assert n == 2


def test_emulated_hist16_wl_small_windows():
    """The 9 728-bin instantiation of pl_hist16_wl on the emulated device: a process of its own whose emulated chip has two CUs
    (HIPEMU_CU_COUNT), so that three frames are "more frames than CUs"."""
    import os
    import subprocess
    import sys

    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from emu_backend import emulated_device\nimport next_row_checks as checks\n"
            "with emulated_device():\n    print('checked', checks.check_hist16_wl_many_frames(torch.device('cuda:0'), n=3))\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, HIPEMU_CU_COUNT="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "checked 6" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_emulated_circle_profile_ring(emulated):
    """pl_circle_profile_ring (LDS-staged annulus) == pl_circle_profile_combined_ex, samples and margins, on the emulated device."""
    import next_row_checks as checks

    assert checks.check_circle_profile_ring(emulated, light=True) == 36


def test_emulated_starshot_batch(golden, emulated):
    """starshot.analyze_batch on the emulated device: the golden "inverted" frame + one shifted copy against the reference's
    numbers and the class API, and the status codes (the four-frame set runs with -m gpu)."""
    import next_row_checks as checks

    assert checks.check_starshot_batch(golden("starshot"), emulated, names=("inverted",), variants=2) == 2


def test_emulated_contrast_rois(golden, emulated):
    import next_row_checks as checks

    checks.check_contrast_rois(golden, emulated)


def test_emulated_canny_integer_images(golden, emulated):
    import next_row_checks as checks

    checks.check_canny_integer_images(golden, emulated)


def test_emulated_rescale_dicom_values(emulated):
    import next_row_checks as checks

    checks.check_rescale_dicom_values(emulated)


def test_emu_stencils_on_awkward_shapes(emu):
    """Gaussian / median / Sobel on frames smaller than the stencil, one-row / one-column frames and odd widths (halos
    that reflect more than once; the packed kernels' shape guards): scipy-exact for every shape."""
    from scipy import ndimage

    rng = np.random.default_rng(12)
    shapes = [(1, 1), (1, 7), (7, 1), (2, 3), (3, 2), (5, 64), (64, 5), (9, 33), (17, 66), (33, 130), (40, 2)]
    for h, w in shapes:
        for dt, code in ((np.uint16, PL_U16), (np.float64, PL_F64), (np.int16, PL_I16)):
            if dt == np.float64:
                f = rng.normal(size=(2, h, w))
            else:
                info = np.iinfo(dt)
                f = rng.integers(info.min, info.max, (2, h, w)).astype(dt)
            for sigma in (1, 2, 5):
                radius = int(4.0 * sigma + 0.5)
                x = np.arange(-radius, radius + 1)
                wts = np.exp(-0.5 / (sigma * sigma) * x ** 2)
                wts = np.ascontiguousarray(wts / wts.sum())
                out, tmp = np.empty_like(f), np.empty_like(f)
                _ok(emu, emu.pl_gaussian2d(_p(f), _p(out), _p(tmp), code, 2, h, w, _p(wts), _p(wts), radius, None))
                want = np.stack([ndimage.gaussian_filter(a, sigma) for a in f])
                np.testing.assert_array_equal(out, want, err_msg=f"gaussian {dt.__name__} {h}x{w} sigma {sigma}")
            for size in (3, 5):
                out = np.empty_like(f)
                _ok(emu, emu.pl_median2d(_p(f), _p(out), code, 2, h, w, size, None))
                want = np.stack([ndimage.median_filter(a, size=size) for a in f])
                np.testing.assert_array_equal(out, want, err_msg=f"median {dt.__name__} {h}x{w} size {size}")
            if dt == np.float64:
                for axis in (0, 1):
                    out = np.empty_like(f)
                    _ok(emu, emu.pl_sobel(_p(f), _p(out), code, 2, h, w, axis, None))
                    want = np.stack([ndimage.sobel(a, axis=axis) for a in f])
                    np.testing.assert_array_equal(out, want, err_msg=f"sobel {h}x{w} axis {axis}")


def test_emulated_find_peaks_sweep(emulated):
    """pl_find_peaks through the host layer on short, flat, plateau-rich and noisy profiles with the argument
    combinations the analyzers use: indices exact, properties to 1e-12 of scipy.signal.find_peaks (the oracle calls it
    as the reference does).  Profiles whose peaks tie exactly on the sort key are skipped (np.argsort's tie order)."""
    from pylinac_amd import profile

    rng = np.random.default_rng(21)
    profs = [np.array([0.0, 1, 0]), np.array([0.0, 1, 1, 0]), np.array([1.0, 1, 1, 1]), np.array([0.0, 2, 1, 2, 0, 3, 0]),
             np.arange(9.0), np.array([0.0, 1, 2, 3, 4, 3, 2, 1, 0])]
    for L in (5, 12, 63, 64, 65, 200, 1030):
        for _ in range(3):
            profs.append(np.round(rng.random(L) * 20) / 4)                       # plateaus and ties
            x = np.linspace(0, 1, L)
            profs.append(sum(np.exp(-0.5 * ((x - c) / 0.03) ** 2) * a for c, a in ((0.2, 1.0), (0.5, 0.7), (0.8, 1.3)))
                         + rng.normal(0, 0.01, L))
    rng2 = np.random.default_rng(22)          # either side of the one-wave-per-profile limit (128 samples), no exact ties
    for L in (127, 128, 129):
        x = np.linspace(0, 1, L)
        profs.append(rng2.random(L))
        profs.append(sum(np.exp(-0.5 * ((x - c) / 0.03) ** 2) * a for c, a in ((0.2, 1.0), (0.5, 0.7), (0.8, 1.3)))
                     + rng2.normal(0, 0.01, L))
    combos = [dict(), dict(threshold=0.3, peak_separation=0.05), dict(threshold=0.5, peak_separation=3, max_number=2),
              dict(fwxm_height=0.8, max_number=1), dict(threshold=0.1, required_prominence=0.2, peak_sort="peak_heights"),
              dict(search_region=(0.2, 0.9), threshold=0.2), dict(min_width=2, peak_separation=0.02)]
    checked = 0
    for v in profs:
        for kw in combos:
            try:
                widx, wprops = orc.find_peaks(v, **kw)
            except (IndexError, ValueError) as exc:
                with pytest.raises(type(exc)):
                    profile.find_peaks(v, **kw)
                continue
            key = wprops[kw.get("peak_sort", "prominences")]
            if len(np.unique(key)) != len(key) or len(np.unique(wprops["peak_heights"])) != len(key):
                continue
            idx, props = profile.find_peaks(v, **kw)
            np.testing.assert_array_equal(idx, widx, err_msg=f"{len(v)} {kw}")
            for k in ("peak_heights", "prominences", "left_ips", "right_ips", "widths", "width_heights"):
                np.testing.assert_allclose(props[k], wprops[k], rtol=1e-12, atol=1e-12, err_msg=f"{k} {len(v)} {kw}")
            for k in ("left_bases", "right_bases"):
                np.testing.assert_array_equal(props[k], wprops[k], err_msg=f"{k} {len(v)} {kw}")
            checked += 1
    assert checked > 150
    # the FWXM search (max_number = 1 by prominence, no filter) takes a short cut through the search; with a prominence bound of
    # zero -- which no peak fails -- the same call takes the general path: same peak, exact ties on the prominence included
    same = 0
    for v in [p for p in profs if len(p) <= 200] + [profs[-1]]:
        for kw in (dict(max_number=1), dict(max_number=1, fwxm_height=0.3, peak_separation=2)):
            a_idx, a = profile.find_peaks(v, **kw)
            b_idx, b = profile.find_peaks(v, required_prominence=0.0, **kw)
            np.testing.assert_array_equal(a_idx, b_idx, err_msg=f"{len(v)} {kw}")
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k} {len(v)} {kw}")
            same += len(a_idx)
    assert same > 50


def test_emulated_thickness_roi(golden, emulated):
    import next_row_checks as checks

    checks.check_thickness_roi(golden, emulated)


def test_emulated_field_strips(golden, emulated):
    import next_row_checks as checks

    checks.check_field_strips(golden, emulated)


def test_emulated_edge_profiles(golden, emulated):
    """InflectionDerivativeProfile / HillProfile on the emulated device (a subset; the full set runs with -m gpu)."""
    import warnings

    import next_row_checks as checks
    from pylinac_amd import profile

    def make(kind, values, **kw):
        cls = profile.HillProfile if kind == "hill" else profile.InflectionDerivativeProfile
        return cls(values, **kw)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        n = checks.check_edge_profiles(golden("edge_profiles"), make,
                                       only=lambda t: t.startswith(("fx2.", "fx4.", "fx11.", "epid.", "fff1.")))
    assert n >= 10
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        checks.check_edge_profile_known_answers(make)


def test_emulated_catphan_volume(golden, emulated):
    import next_row_checks as checks

    checks.check_catphan_volume(golden, emulated, names=("b",))


def test_emu_label_matches_scipy(emu):
    """Union-find labelling (default per-pixel start) == scipy.ndimage.label numbering, 4- and 8-connected."""
    _check_label(emu)


def test_emu_fill_holes_and_clear_border(emu):
    """The two other users of the union-find roots: binary_fill_holes (background labelling) and clear_border."""
    _check_fill_holes_and_clear_border(emu)


def test_emulated_profile_base_fields(golden, emulated):
    import next_row_checks as checks

    checks.check_profile_base_fields(golden("edge_profiles"))


def test_emulated_as_resampled(emulated):
    import next_row_checks as checks

    checks.check_as_resampled(emulated)


def test_emulated_reductions_statistics_sweep(emulated):
    """Axis reductions, min/max, Otsu, percentiles and circle profiles on small odd shapes and several dtypes against
    numpy / the oracle (shapes the `-m gpu` parity tests do not visit)."""
    import torch

    from pylinac_amd import ops

    rng = np.random.default_rng(33)
    dev = emulated
    for h, w in ((1, 1), (1, 9), (9, 1), (3, 65), (65, 3), (17, 31), (40, 128)):
        for dt in (np.uint16, np.int16, np.float32, np.float64, np.uint8):
            if np.issubdtype(dt, np.integer):
                info = np.iinfo(dt)
                f = rng.integers(info.min, info.max, (2, h, w)).astype(dt)
            else:
                f = (rng.normal(size=(2, h, w)) * 50).astype(dt)
            t = torch.from_numpy(f).to(dev)
            mn, mx = ops.minmax(t)
            assert np.array_equal(mn.cpu().numpy(), f.reshape(2, -1).min(1)) and np.array_equal(mx.cpu().numpy(), f.reshape(2, -1).max(1))
            for axis in (0, 1):
                for op, fn in (("sum", np.sum), ("mean", np.mean), ("max", np.max), ("min", np.min)):
                    got = ops.reduce_axis(t, axis, op).cpu().numpy()
                    want = fn(f if dt == np.float32 else f.astype(np.float64), axis=axis + 1)
                    if np.issubdtype(dt, np.integer) or op in ("max", "min"):
                        assert np.array_equal(got, want), (dt, h, w, axis, op)
                    elif dt == np.float32:      # numpy accumulates float32 sums in float32 (pairwise): a few ulps
                        assert np.allclose(got, want, rtol=2e-6, atol=1e-4), (dt, h, w, axis, op)
                    else:
                        assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (dt, h, w, axis, op)
            if dt in (np.uint16, np.int16) and h * w >= 9:
                thr = ops.threshold_otsu(t).cpu().numpy()
                want = np.array([orc.threshold_otsu(a) for a in f])
                assert np.array_equal(thr, want), (dt, h, w, thr, want)
                qs = [0, 5, 37.5, 50, 99.9, 100]
                got = ops.percentile(t, qs).cpu().numpy()
                assert np.array_equal(got, np.stack([np.percentile(a, qs) for a in f])), (dt, h, w)
    # circle profiles: centres near / outside the border, radii that leave the frame (out-of-bounds samples read 0)
    img = rng.integers(0, 60000, (1, 37, 53)).astype(np.uint16)
    t = torch.from_numpy(img).to(dev)
    for cx, cy, r in ((26.2, 18.7, 9.5), (3.0, 4.0, 7.0), (50.5, 35.5, 11.0), (-4.0, 10.0, 8.0), (26.0, 18.0, 40.0)):
        size = np.pi * r * 2
        got = ops.circle_profile(t, cx, cy, [r], size)[0].cpu().numpy()
        want = orc.circle_profile(img[0], (cx, cy), r).astype(np.float64)
        assert np.array_equal(got, want), (cx, cy, r)
        radii = np.linspace(r * 0.9, r * 1.1, 5)
        got = ops.circle_profile(t, cx, cy, radii, np.pi * radii.max() * 2, 0, True, 5.0)[0].cpu().numpy()
        want = orc.collapsed_circle_profile(img[0], (cx, cy), r, width_ratio=0.1, num_profiles=5)
        assert np.allclose(got, want, rtol=1e-13, atol=1e-9), (cx, cy, r)


def test_emulated_bb_finder_random_windows(emulated):
    """find_features_batch (label -> region table -> pl_features_level: flood fill, perimeter, hull, predicates, weighted
    centroid, de-duplication) on randomly generated BB windows -- sizes, positions, blur, noise, a second BB, a rod --
    against the oracle's restatement of the reference's find_features (itself pinned to the reference under scikit-image
    0.18.3): same level, same count, centroids to 1e-10."""
    import torch
    from scipy import ndimage

    from pylinac_amd import features as pf

    rng = np.random.default_rng(77)
    dpmm = 2.98
    wins, specs = [], []
    for k in range(3):          # k = 1: a second BB; k = 2: a rod (the emulator runs a window in ~10 s; -m gpu runs hundreds)
        n = int(rng.integers(90, 150))
        yy, xx = np.mgrid[0:n, 0:n].astype(float)
        img = np.full((n, n), 0.2)
        r_mm = 2.5 * rng.uniform(0.85, 1.15)
        cy, cx = rng.uniform(n * 0.3, n * 0.7, 2)
        img[np.hypot(yy - cy, xx - cx) < r_mm * dpmm] = 1.0
        if k % 3 == 1:          # a second BB, well separated
            cy2, cx2 = cy + rng.choice([-1, 1]) * 30, cx + rng.choice([-1, 1]) * 25
            img[np.hypot(yy - cy2, xx - cx2) < 2.5 * dpmm] = 0.9
        if k % 4 == 2:          # a thin rod attached to the BB
            img[int(cy) - 1:int(cy) + 1, int(cx):min(n - 8, int(cx) + 40)] = 0.8
        img = ndimage.gaussian_filter(img, rng.uniform(0.6, 1.6)) + rng.normal(0, 0.01, (n, n))
        wins.append(img)
        specs.append((2 if k % 3 == 1 else 1, 5.0))
    checked = 0
    for img, (maxn, minsep) in zip(wins, specs):
        try:
            ref_pts, ref_level = orc.find_features_restated(img, dpmm, 2.5, 0.5, max_number=maxn, min_separation_mm=minsep)
        except ValueError:
            ref_pts, ref_level = [], -1
        res = pf.find_features_batch(torch.from_numpy(img[None]).to(emulated), dpmm, 2.5, 0.5, max_number=maxn,
                                     min_separation_mm=minsep)
        assert int(res["status"][0]) == 0
        assert int(res["count"][0]) == len(ref_pts) and int(res["level"][0]) == ref_level, (len(ref_pts), ref_level)
        if ref_pts:
            got = res["xy"][0, : len(ref_pts)].cpu().numpy()
            assert np.allclose(got, np.array(ref_pts), rtol=1e-10, atol=1e-10)
            checked += 1
        # the level-by-level path (the fallback of the one-launch sweep) gives the same answer
        lv = pf.find_features_batch(torch.from_numpy(img[None]).to(emulated), dpmm, 2.5, 0.5, max_number=maxn,
                                    min_separation_mm=minsep, level_by_level=True)
        assert int(lv["count"][0]) == int(res["count"][0]) and int(lv["level"][0]) == int(res["level"][0])
        assert np.array_equal(lv["xy"][0, : len(ref_pts)].cpu().numpy(), res["xy"][0, : len(ref_pts)].cpu().numpy())
    assert checked >= 2


def test_emulated_picket_fence_random_frames(emulated):
    """analyze_batch (scaled column mean, picket peaks, leaf x picket windows, FWXM positions) on random picket-fence
    frames -- picket count, spacing, gap, pixel size, frame size, hidden rows -- against the oracle's per-image loop
    (pinned to the reference's own PicketFence.analyze): same NaN pattern, identical positions."""
    import torch

    from pylinac_amd import picketfence as ppf
    from tests.golden.make_golden import pf_frame

    rng = np.random.default_rng(88)
    for k in range(6):
        pixel = float(rng.choice([0.78125, 0.6, 1.0]))
        h, w = int(rng.integers(120, 200)), int(rng.integers(260, 420))
        n_pickets = int(rng.integers(3, 8))
        spacing = float(rng.uniform(18, 30))
        frame = pf_frame(h + 8, w + 8, pixel, 4000 + k, n_pickets=n_pickets, spacing_mm=spacing,
                         gap_mm=float(rng.uniform(1.5, 3.5)))[4:-4, 4:-4]
        frame = np.ascontiguousarray(frame)
        res = ppf.analyze_batch(torch.from_numpy(frame[None]).to(emulated), 1 / pixel)
        ref = orc.pf_measure(orc.normalize(orc.ground(frame)), 1 / pixel)
        P = len(ref["peak_idxs"])
        assert int(res.picket_count[0]) == P, (k, P)
        pos = res.position[0, :, :P].cpu().numpy()
        assert np.array_equal(np.isnan(pos), np.isnan(ref["position"])), k
        assert np.array_equal(pos[~np.isnan(pos)], ref["position"][~np.isnan(pos)]), k
    # leaf heights either side of the column-median networks: MLCi leaves (10 mm) at 0.4 mm -> 25 rows (32-value network),
    # at 0.26 mm -> 38 rows (beyond the networks: rank counting), Millennium at 0.8 mm -> 6 / 12 rows (16-value network)
    for k, (pixel, mlc, h, w) in enumerate([(0.4, "MLCI", 150, 300), (0.26, "MLCI", 160, 360), (0.8, "MILLENNIUM", 130, 280)]):
        frame = pf_frame(h + 8, w + 8, pixel, 4100 + k, n_pickets=4, spacing_mm=w * pixel / 6, gap_mm=3.0)[4:-4, 4:-4]
        frame = np.ascontiguousarray(frame)
        res = ppf.analyze_batch(torch.from_numpy(frame[None]).to(emulated), 1 / pixel, mlc=mlc)
        ref = orc.pf_measure(orc.normalize(orc.ground(frame)), 1 / pixel, mlc=mlc)
        P = len(ref["peak_idxs"])
        assert int(res.picket_count[0]) == P and P >= 3, (k, P)
        pos = res.position[0, :, :P].cpu().numpy()
        assert np.isfinite(pos).sum() >= P, k
        assert np.array_equal(np.isnan(pos), np.isnan(ref["position"])), k
        assert np.array_equal(pos[~np.isnan(pos)], ref["position"][~np.isnan(pos)]), k


def test_emulated_wl_field_xim_gamma_random(emulated):
    """Three more randomised audits on the emulated device: the WL field CAX (percentile threshold -> fill holes ->
    centre of mass) on random field / BB geometries, XIM decoding of random images whose differences need 1-, 2- and
    4-byte codes, and gamma_2d on random dose pairs -- all against the pinned restatements, exact."""
    import torch

    from pylinac_amd import gamma as pg
    from pylinac_amd import winston_lutz, xim

    rng = np.random.default_rng(99)
    # ---- WL field CAX
    frames = []
    for k in range(5):
        h, w = int(rng.integers(90, 160)), int(rng.integers(90, 160))
        yy, xx = np.mgrid[0:h, 0:w]
        cy, cx, half = rng.integers(35, h - 35), rng.integers(35, w - 35), int(rng.integers(10, 25))
        img = np.where((abs(yy - cy) < half) & (abs(xx - cx) < half), 42000, 1500).astype(np.int64)
        bb = ((yy - cy - rng.integers(-4, 5)) ** 2 + (xx - cx - rng.integers(-4, 5)) ** 2) < rng.integers(9, 40)
        img[bb] = 11000
        img = img + rng.integers(0, 300, img.shape)
        frames.append(img.astype(np.uint16))
    for f in frames:
        got = winston_lutz.field_centroids_batch(torch.from_numpy(f[None]).to(emulated))[0].cpu().numpy()
        want = orc.wl_field_centroid(f)
        assert np.allclose(got, want, rtol=0, atol=1e-9), (got, want)
    # ---- XIM
    for k, (h, w, step) in enumerate(((23, 31, 90), (40, 17, 20000), (9, 64, 3000000))):
        img = np.cumsum(rng.integers(-step, step, (h, w)), axis=1)
        img = (img + rng.integers(-step, step, (h, 1))).astype(np.int32)
        lut, stream = orc.xim_encode(img)
        out = xim.decode_xim_pixels(lut, stream, w, h, 4, device=emulated).cpu().numpy()
        assert np.array_equal(out, img), k
    # ---- gamma_2d
    for k in range(3):
        ref = rng.uniform(0, 100, (20 + k, 24)) * (rng.random((20 + k, 24)) > 0.05)
        ev = ref * rng.uniform(0.95, 1.05, ref.shape)
        kw = dict(dose_to_agreement=float(rng.choice([1, 2, 3])), distance_to_agreement=int(rng.integers(1, 4)),
                  gamma_cap_value=2, global_dose=bool(k % 2), dose_threshold=float(rng.choice([0, 5, 10])))
        got = pg.gamma_2d(ref, ev, device=emulated, **kw).cpu().numpy()
        want = orc.gamma_2d(ref, ev, **kw)
        assert np.array_equal(got, want, equal_nan=True), kw


def test_emulated_bit_invert_and_convert_to_dtype(emulated):
    import next_row_checks as checks

    checks.check_bit_invert_and_convert_to_dtype()


def test_emulated_rotate(emulated, golden):
    import next_row_checks as checks

    checks.check_rotate(golden, emulated)


def test_emulated_large_rois(emulated):
    import next_row_checks as checks

    checks.check_large_rois(emulated)


def test_emulated_mask_regions_vs_scipy(emulated):
    """slice_regions.hip: bit-plane / row-run labelling of clear_border -> fill_holes -> label -> regionprops in one workgroup."""
    import next_row_checks as checks

    checked, overflowed = checks.check_mask_regions(emulated)
    assert checked >= 150


def test_emulated_scharr_gaussian_bit_identical(emulated):
    import next_row_checks as checks

    checks.check_scharr_gaussian(emulated)


def test_emulated_picket_fence_left_right_and_separate_leaves(golden, emulated):
    import next_row_checks as checks

    checks.check_pf_orientation_device(golden("picketfence_orient"), emulated)


def test_emulated_ground_promotion(emulated):
    import next_row_checks as checks

    checks.check_ground_promotion()


def test_emulated_edge_otsu_one_launch(emulated):
    import next_row_checks as checks

    checks.check_edge_otsu(emulated, shapes=((2, 70, 130, np.int16), (1, 33, 65, np.uint16)), sigmas=(1,))


def test_emulated_circle_profile_combined(emulated):
    import next_row_checks as checks

    checks.check_circle_profile_combined(emulated)


def test_emulated_phantom_roi_fused_vs_separate(emulated):
    import next_row_checks as checks

    checks.check_phantom_roi_fused_vs_separate(emulated, slices=(44,))


def test_emulated_histogram16_one_read(emulated):
    """The single-read two-window histogram (frames of >= 2^18 pixels) == np.bincount on the emulated device."""
    import next_row_checks as checks

    assert checks.check_histogram16_one_read(emulated, sizes=((512, 512), (513, 520))) == 32


def test_emulated_fused_tail_vs_separate(emulated):
    """The pipeline's per-band column sums + one-launch tail == the separate colsum / mean / find_peaks / fwxm_record launches."""
    import next_row_checks as checks

    assert checks.check_fused_tail_vs_separate(emulated) == 4


def test_emulated_canny_masked(golden, emulated):
    import next_row_checks as checks

    checks.check_canny_masked(golden, emulated)


def test_emulated_gamma_geometric(golden, emulated):
    import next_row_checks as checks

    checks.check_gamma_geometric(golden, emulated)


def test_emulated_picket_fence_other_leaf_banks(golden, emulated):
    import next_row_checks as checks

    checks.check_pf_mlc_device(golden("picketfence_mlc"), emulated)


def test_emulated_fwxm_short_profiles(emulated):
    import next_row_checks as checks

    checks.check_fwxm_short_profiles(emulated, trials=240)


def test_emulated_field_cax_tile_maxima(emulated):
    import next_row_checks as checks

    checks.check_field_cax_tile_maxima(emulated)


def test_emulated_bb_sweep_run_table_tiers(emulated):
    import next_row_checks as checks

    assert checks.check_bb_sweep_run_table_tiers(emulated, full=False) == 2


def test_emulated_pack_columns(emulated):
    """pl_pack_columns: float64 / int32 columns, strided sources with offsets and additive constants -> one float64 table."""
    import torch

    from pylinac_amd import ops

    n = 37
    a = torch.arange(n * 3, dtype=torch.float64, device=emulated).reshape(n, 3) * 0.5
    b = torch.arange(n * 8 * 2, dtype=torch.float64, device=emulated).reshape(n, 8, 2) - 7.25
    c = torch.arange(n, dtype=torch.int32, device=emulated) * -3
    got = ops.pack_columns([(a, 1, 0.0), (a, 0, 2.5), (b, 0, 10.0), (b, 1, -1.0), c, (c, 0, 0.5)], n).cpu().numpy()
    want = np.stack([a[:, 1].numpy(), a[:, 0].numpy() + 2.5, b[:, 0, 0].numpy() + 10.0, b[:, 0, 1].numpy() - 1.0,
                     c.numpy().astype(float), c.numpy() + 0.5], axis=1)
    assert np.array_equal(got, want)
    with pytest.raises(TypeError):
        ops.pack_columns([torch.zeros(n, dtype=torch.float32, device=emulated)], n)
    with pytest.raises(ValueError):
        ops.pack_columns([a] * 17, n)


def test_emulated_dicom_decode(golden, emulated):
    """f1, the DICOM half: the Part-10 walk + pl_dicom_decode on the emulated device against the fixtures and np.frombuffer."""
    import next_row_checks as checks

    checks.check_dicom_golden(golden, emulated)
    checks.check_dicom_decode_fuzz(emulated, frame_shapes=((6, 10),), n=2)


def test_emulated_ctp528_device_axis_path(emulated):
    import next_row_checks as checks

    checks.check_ctp528_device_axis_path(emulated, n_slices=6, light=True)


def test_emulated_wl_analyze_batch_other_dtypes(golden, emulated):
    import next_row_checks as checks

    checks.check_wl_analyze_batch_other_dtypes(golden, emulated, frames=(0, 7), int16_frames=(0,))


def test_emulated_pf_other_dtypes(golden, emulated):
    import next_row_checks as checks

    checks.check_pf_other_dtypes(golden("picketfence_mlc"), emulated)


def test_emulated_edge_plane32(emulated):
    """The packed-float32 edge kernel and its bracketed consumers against the exact path (small frames; the GPU suite adds
    CatPhan slices)."""
    import next_row_checks as checks

    worst = checks.check_edge_plane32(emulated, shapes=((2, 70, 130, np.int16), (1, 40, 66, np.uint16)))
    assert worst <= 32
