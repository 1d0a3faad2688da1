"""GPU (-m gpu): parity of the HIP path -- called through the C ABI -- against
  * the committed golden vectors produced by the reference's own code (tests/golden/),
  * the CPU oracle on seeded inputs, and
  * size-independent properties at BASELINE's full 1024x1024 size.
Bars: bit-exact for integer frames / thresholds / indices; float results compared with
np.array_equal where the reference's operation order is defined, otherwise rtol 1e-12
(north_star asks 1e-5)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pylinac_oracle as o

pytestmark = pytest.mark.gpu

FILTERS = [("g5", 5, "gaussian"), ("g1", 1, "gaussian"), ("g2", 2, "gaussian"), ("gf03", 0.03, "gaussian"),
           ("m3", 3, "median"), ("m5", 5, "median"), ("m2", 2, "median"), ("mf05", 0.05, "median")]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_native_library_is_loaded(dev):
    from pylinac_amd import _lib

    lib = _lib.load()
    assert lib.pl_device_available() == 1
    maps = open("/proc/self/maps").read()
    assert "libpylinac_hip.so" in maps


# ------------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name", ["field", "random", "tiny"])
def test_filters_vs_reference_golden(golden, dev, name):
    from pylinac_amd import array_utils as au
    from pylinac_amd.image import ArrayImage, ImageBatch

    g = golden("frames")
    fs = g[f"{name}.in"]
    for tag, size, kind in FILTERS:
        ref = g[f"{name}.filter.{tag}"]
        # batched device path
        b = ImageBatch(T(fs, dev))
        b.filter(size, kind)
        got = b.array.cpu().numpy()
        assert got.dtype == ref.dtype and np.array_equal(got, ref), (name, tag, "batch")
        # numpy drop-in path (ArrayImage.filter, pylinac/core/image.py:695)
        im = ArrayImage(fs[0].copy())
        im.filter(size=size, kind=kind)
        assert np.array_equal(im.array, ref[0]), (name, tag, "ArrayImage")
        assert np.array_equal(au.filter(fs[-1], size, kind), ref[-1]), (name, tag, "array_utils")


@pytest.mark.parametrize("name", ["field", "random", "tiny"])
def test_mutators_vs_reference_golden(golden, dev, name):
    from pylinac_amd import array_utils as au
    from pylinac_amd import ops
    from pylinac_amd.image import ArrayImage

    g = golden("frames")
    fs = g[f"{name}.in"]
    t = T(fs, dev)
    assert np.array_equal(ops.threshold(t, 30000).cpu().numpy(), g[f"{name}.threshold.high"])
    assert np.array_equal(ops.threshold(t, 30000, "low").cpu().numpy(), g[f"{name}.threshold.low"])
    assert np.array_equal(ops.as_binary(t, 30000).cpu().numpy(), g[f"{name}.as_binary"])
    assert np.array_equal(ops.ground(t).cpu().numpy(), g[f"{name}.ground"])
    assert np.array_equal(ops.normalize(t).cpu().numpy(), g[f"{name}.normalize"])
    assert np.array_equal(ops.invert(t).cpu().numpy(), g[f"{name}.invert"])
    assert np.array_equal(ops.percentile(t, [0.5, 5, 50, 95, 99.5, 99.9]).numpy(), g[f"{name}.percentiles"])
    for i, f in enumerate(fs):
        assert np.array_equal(au.stretch(f.astype(float), 0, 1), g[f"{name}.stretch"][i])
        im = ArrayImage(f.copy())
        im.threshold(30000)
        assert im.array.dtype == f.dtype and np.array_equal(im.array, g[f"{name}.threshold.high"][i])
        b = ArrayImage(f.copy()).as_binary(30000)
        assert b.array.dtype == np.int64 and np.array_equal(b.array, g[f"{name}.as_binary"][i])
        im = ArrayImage(f.copy())
        mn = im.ground()
        assert mn == f.min() and np.array_equal(im.array, g[f"{name}.ground"][i])
        im = ArrayImage(f.copy())
        im.normalize()
        assert im.array.dtype == np.float64 and np.array_equal(im.array, g[f"{name}.normalize"][i])
        im = ArrayImage(f.copy())
        im.invert()
        assert np.array_equal(im.array, g[f"{name}.invert"][i])


def test_float_and_int16_filters_vs_golden(golden, dev):
    from pylinac_amd import array_utils as au

    g = golden("frames")
    for key, size, kind in [("float64.filter.g2", 2, "gaussian"), ("float64.filter.m3", 3, "median"),
                            ("float32.filter.g2", 2, "gaussian"), ("int16.filter.g2", 2, "gaussian"),
                            ("int16.filter.m3", 3, "median")]:
        fs = g[key.split(".")[0] + ".in"]
        ref = g[key]
        got = np.stack([au.filter(f, size, kind) for f in fs])
        assert got.dtype == ref.dtype and np.array_equal(got, ref), key


def test_reference_known_answer_tests_on_gpu(golden, dev):
    """The reference's own KATs (tests_basic/core/test_array_utils.py:65-149,
    tests_basic/core/test_image.py:450-461, 522-535) through the drop-in API."""
    from pylinac_amd import array_utils as au
    from pylinac_amd.image import ArrayImage

    kat = np.array([0, 0, 0, 3, 0, 0, 0])
    assert np.array_equal(au.filter(kat, size=1, kind="median"), [0, 0, 0, 3, 0, 0, 0])
    assert np.array_equal(au.filter(kat, size=0.1, kind="median"), [0, 0, 0, 3, 0, 0, 0])
    assert np.array_equal(au.filter(kat, size=3, kind="median"), [0, 0, 0, 0, 0, 0, 0])
    assert np.array_equal(au.filter(np.array([0, 0, 3, 3, 0, 0, 0]), size=3, kind="median"), [0, 0, 3, 3, 0, 0, 0])
    assert np.array_equal(au.filter(kat, size=1, kind="gaussian"), [0, 0, 0, 1, 0, 0, 0])
    with pytest.raises(ValueError):
        au.filter(kat, size=2.3, kind="gaussian")
    with pytest.raises(ValueError):
        au.filter(kat, size=1, kind="filterthis")
    gf = golden("frames")
    im = ArrayImage(np.arange(42).reshape(6, 7))
    im.filter(3)
    assert im.array[0, 0] == 1 and np.array_equal(im.array, gf["kat.image.filter3"])
    im = ArrayImage(np.arange(42).reshape(6, 7))
    im.threshold(10)
    assert im.array[0, 4] == 0 and np.array_equal(im.array, gf["kat.image.threshold10"])
    im = ArrayImage(np.arange(42).reshape(6, 7))
    im.threshold(20, kind="low")
    assert np.array_equal(im.array, gf["kat.image.threshold20low"])
    n = au.normalize(np.array((1, 2, 3, 4)))
    assert n.max() == 1.0 and n[-1] == 1.0 and n[0] == 0.25
    assert np.array_equal(au.normalize(np.array((1, 2, 3, 4), dtype=float), 2), [0.5, 1, 1.5, 2])
    assert np.array_equal(au.invert(np.array([0, 10])), [10, 0])
    assert np.array_equal(au.invert(np.array([-5, -1])), [-1, -5])
    assert np.array_equal(au.ground(np.array([3, 4, 5])), [0, 1, 2])
    assert np.array_equal(au.ground(np.array([-3, -4, -5])), [2, 1, 0])
    assert np.array_equal(au.ground(np.array([3, 4, 5]), value=10), [10, 11, 12])


def test_otsu_vs_skimage_golden(golden, dev):
    from pylinac_amd import ops

    g = golden("otsu")
    for k in ["u16_field", "u16_random", "i16", "const", "two_level"]:
        got = ops.threshold_otsu(T(g[f"{k}.in"], dev)).cpu().numpy()
        assert np.array_equal(got, g[f"{k}.otsu"]), k


def test_otsu16_single_pass(golden, dev):
    """a4 (fast path): pl_otsu16 (LDS-window histogram + Otsu in one read of the frame; frames that do not fit the window
    -- full-range noise, outliers the row sample missed -- take the gated two-kernel path inside the same call)."""
    import next_row_checks as checks

    checks.check_otsu16(golden, dev)


PROFILES = ["simple9", "simple8", "long23", "long22", "skewed19", "sigmoid21", "sawtooth", "walk600", "pickets",
            "noisy_field"]


def test_find_peaks_vs_reference_golden(golden, dev):
    from pylinac_amd import profile as pp

    g = golden("peaks")
    variants = json.loads(str(g["variants"]))
    checked = 0
    for pname in PROFILES:
        vals = g[f"{pname}.values"]
        for vname, kw in variants.items():
            kw = dict(kw)
            if "search_region" in kw:
                kw["search_region"] = tuple(kw["search_region"])
            key = f"{pname}.{vname}"
            if f"{key}.error" in g.files:
                continue
            idx, props = pp.find_peaks(vals, **kw)
            assert np.array_equal(idx, g[f"{key}.idx"]), key
            for k in ("peak_heights", "prominences", "left_bases", "right_bases", "widths", "width_heights",
                      "left_ips", "right_ips"):
                assert np.array_equal(props[k], g[f"{key}.{k}"]), (key, k)
            checked += 1
    assert checked > 50


def test_multiprofile_and_fwxm_vs_reference_golden(golden, dev):
    from pylinac_amd.profile import FWXMProfile, MultiProfile

    g = golden("peaks")
    for pname in PROFILES:
        vals = g[f"{pname}.values"]
        mp = MultiProfile(vals.copy())
        for tag, fn in [("peaks", mp.find_peaks), ("valleys", mp.find_valleys), ("fwxm", mp.find_fwxm_peaks)]:
            i, v = fn()
            assert np.array_equal(i, g[f"{pname}.mp.{tag}.idx"]), (pname, tag)
            assert np.array_equal(v, g[f"{pname}.mp.{tag}.val"]), (pname, tag)
        for h in (25, 50, 75):
            if f"{pname}.fwxm{h}.error" in g.files:
                with pytest.raises(IndexError):
                    FWXMProfile(vals.copy(), fwxm_height=h).field_edge_idx("left")
            else:
                fp = FWXMProfile(vals.copy(), fwxm_height=h)
                got = np.array([fp.field_edge_idx("left"), fp.field_edge_idx("right"), fp.center_idx, fp.field_width_px])
                assert np.array_equal(got, g[f"{pname}.fwxm{h}"]), (pname, h)


def test_fwxm_known_answers_on_gpu(dev):
    """tests_basic/core/test_profile.py:272-325."""
    from pylinac_amd.profile import FWXMProfile

    s9 = np.array([0, 1, 2, 3, 4, 3, 2, 1, 0], dtype=float)
    s8 = np.array([0, 1, 2, 3, 3, 2, 1, 0], dtype=float)
    sk = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 8, 6, 4, 2, 0], dtype=float)
    p = FWXMProfile(s9)
    assert (p.field_edge_idx("left"), p.field_edge_idx("right"), p.center_idx, p.field_width_px) == (2, 6, 4, 4)
    p = FWXMProfile(s8)
    assert (p.field_edge_idx("left"), p.field_edge_idx("right"), p.center_idx) == (1.5, 5.5, 3.5)
    p = FWXMProfile(s9, fwxm_height=25)
    assert (p.field_edge_idx("left"), p.field_edge_idx("right"), p.field_width_px) == (1, 7, 6)
    p = FWXMProfile(s9, fwxm_height=75)
    assert (p.field_edge_idx("left"), p.field_edge_idx("right")) == (3, 5)
    p = FWXMProfile(sk)
    assert (p.field_edge_idx("left"), p.field_edge_idx("right"), p.field_width_px) == (5, 14.5, 9.5)
    # x-values that decrease are re-sorted (test_profile.py:216-226)
    p = FWXMProfile(s9, x_values=np.arange(9)[::-1])
    assert p.field_width_px == 4
    x_bad = np.arange(9)
    x_bad[2] = 5
    with pytest.raises(ValueError):
        FWXMProfile(s9, x_values=x_bad)
    with pytest.raises(IndexError):  # no peak at all: same exception as the reference (profile.py:608)
        FWXMProfile(np.zeros(12)).field_edge_idx("left")


def test_pipeline_vs_reference_golden(golden, dev):
    from pylinac_amd.pipeline import EpidPipeline

    g = golden("epid_pipeline")
    fr = g["in"]
    n, h, w = fr.shape
    res = EpidPipeline(n, h, w, dev).run(T(fr, dev))
    assert np.array_equal(res.frames.cpu().numpy(), g["out"])
    assert np.array_equal(res.threshold.cpu().numpy(), g["otsu"])
    assert np.array_equal(res.profile.cpu().numpy(), g["profile"])
    assert np.array_equal(res.fwxm.cpu().numpy()[:, 4:8], g["fwxm"])
    assert int(res.status.abs().sum()) == 0


# ------------------------------------------------------------------- oracle on seeded random inputs
@pytest.mark.parametrize("shape", [(2, 50, 70), (1, 129, 1000), (3, 33, 17), (2, 1, 300), (2, 300, 1), (1, 5, 5)])
def test_filters_vs_oracle_ragged_shapes(dev, shape):
    from pylinac_amd import ops

    rng = np.random.default_rng(sum(shape))
    a = rng.integers(0, 65536, shape, dtype=np.uint16)
    for sigma in (1, 2, 3, 5, 6):  # 6 -> radius 24: generic kernel; others: specialised kernels
        ref = np.stack([o.filter(f, sigma, "gaussian") for f in a])
        assert np.array_equal(ops.gaussian_filter(T(a, dev), sigma).cpu().numpy(), ref), sigma
    for size in (1, 2, 3, 4, 7):
        ref = np.stack([o.filter(f, size, "median") for f in a])
        assert np.array_equal(ops.median_filter(T(a, dev), size).cpu().numpy(), ref), size


def test_gaussian_extreme_values_and_constants(dev):
    """Saturated / constant regions are where the float64 sum sits within 1e-11 of an integer:
    the summation ORDER decides the truncated result."""
    from pylinac_amd import ops

    for v in (0, 1, 255, 4095, 32768, 65534, 65535):
        a = np.full((1, 64, 96), v, dtype=np.uint16)
        for sigma in (1, 2, 5):
            assert np.array_equal(ops.gaussian_filter(T(a, dev), sigma).cpu().numpy()[0], o.filter(a[0], sigma, "gaussian"))
    rng = np.random.default_rng(0)
    a = (rng.integers(0, 2, (2, 80, 120)) * 65535).astype(np.uint16)  # only 0 / 65535
    for sigma in (2, 5):
        ref = np.stack([o.filter(f, sigma, "gaussian") for f in a])
        assert np.array_equal(ops.gaussian_filter(T(a, dev), sigma).cpu().numpy(), ref)


def test_gaussian_fma_decision_and_exact_fallback_mix(dev):
    """Integer frames take a float64-FMA chain whose truncation is provably scipy's unless the sum
    lies within ~1e-9 of an integer; then the exact chain runs.  Frames that mix noisy areas
    (fast path) with constant blocks, ramps and saturated areas (fallback) exercise both in one
    wave, for every specialised radius and for the signed dtypes."""
    from pylinac_amd import ops

    rng = np.random.default_rng(12)
    a = rng.integers(0, 65536, (3, 150, 1100), dtype=np.uint16)
    a[0, 20:90, 100:700] = 2000
    a[0, 60:140, 650:1000] = 65535
    a[1, :, :550] = 0
    a[1, 40:100, 560:900] = np.arange(340, dtype=np.uint16)[None, :] * 3
    a[2, 10:140, :] = (np.arange(130, dtype=np.uint16)[:, None] * 500)
    for sigma in (1, 2, 3, 5):
        ref = np.stack([o.filter(f, sigma, "gaussian") for f in a])
        assert np.array_equal(ops.gaussian_filter(T(a, dev), sigma).cpu().numpy(), ref), sigma
    ai = (a.astype(np.int32) - 32768).astype(np.int16)
    for sigma in (2, 5):
        ref = np.stack([o.filter(f, sigma, "gaussian") for f in ai])
        assert np.array_equal(ops.gaussian_filter(T(ai, dev), sigma).cpu().numpy(), ref), sigma
    a32 = (a.astype(np.int32) - 30000) * 30000   # large magnitudes: bound scales with |sum|
    ref = np.stack([o.filter(f, 2, "gaussian") for f in a32])
    assert np.array_equal(ops.gaussian_filter(T(a32, dev), 2).cpu().numpy(), ref)
    a64 = a.astype(np.int64) * (1 << 40)          # beyond 2^52: every output must take the exact chain
    ref = np.stack([o.filter(f, 2, "gaussian") for f in a64[:1, :40, :300]])
    assert np.array_equal(ops.gaussian_filter(T(a64[:1, :40, :300], dev), 2).cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32, np.int64, np.float32, np.float64])
def test_other_dtypes_vs_oracle(dev, dtype):
    from pylinac_amd import ops

    rng = np.random.default_rng(3)
    if np.dtype(dtype).kind == "f":
        a = rng.normal(100, 30, (2, 40, 56)).astype(dtype)
    else:
        info = np.iinfo(dtype)
        a = rng.integers(max(info.min, -30000), min(info.max, 30000), (2, 40, 56)).astype(dtype)
    t = T(a, dev)
    assert np.array_equal(ops.gaussian_filter(t, 2).cpu().numpy(), np.stack([o.filter(f, 2, "gaussian") for f in a]))
    assert np.array_equal(ops.median_filter(t, 3).cpu().numpy(), np.stack([o.filter(f, 3, "median") for f in a]))
    assert np.array_equal(ops.median_filter(t, 5).cpu().numpy(), np.stack([o.filter(f, 5, "median") for f in a]))
    assert np.array_equal(ops.ground(t).cpu().numpy(), np.stack([o.ground(f) for f in a]))
    assert np.array_equal(ops.invert(t).cpu().numpy(), np.stack([o.invert(f) for f in a]))
    nrm = ops.normalize(t).cpu().numpy()
    ref = np.stack([o.normalize(f) for f in a])
    assert nrm.dtype == ref.dtype and np.array_equal(nrm, ref)
    thr = float(np.median(a))
    assert np.array_equal(ops.threshold(t, thr).cpu().numpy(), o.threshold(a, thr))
    mn, mx = ops.minmax(t)
    assert np.array_equal(mn.cpu().numpy(), a.min(axis=(1, 2)).astype(float))
    assert np.array_equal(mx.cpu().numpy(), a.max(axis=(1, 2)).astype(float))


def test_profiles_reductions_vs_numpy(dev):
    from pylinac_amd import ops
    from pylinac_amd.image import ArrayImage

    rng = np.random.default_rng(8)
    a = rng.integers(0, 65536, (3, 77, 130), dtype=np.uint16)
    for ax in (0, 1):
        for op in ("mean", "sum", "max", "min"):
            ref = getattr(np, op)(a, axis=ax + 1).astype(np.float64)
            assert np.array_equal(ops.reduce_axis(T(a, dev), ax, op).cpu().numpy(), ref), (ax, op)
    im = ArrayImage(a[0])
    assert np.array_equal(im.profile(0, "mean"), np.mean(a[0], 0))
    assert np.array_equal(im.profile(1, "max"), np.max(a[0], 1))
    f = rng.normal(size=(2, 60, 90))
    assert np.array_equal(ops.reduce_axis(T(f, dev), 0, "mean").cpu().numpy(), np.mean(f, axis=1))
    # along the contiguous axis numpy sums pairwise: equal to rounding, not bitwise
    assert np.allclose(ops.reduce_axis(T(f, dev), 1, "mean").cpu().numpy(), np.mean(f, axis=2), rtol=1e-12, atol=1e-15)


def test_histogram_otsu_percentile_vs_oracle(dev):
    from pylinac_amd import ops

    rng = np.random.default_rng(4)
    a = rng.integers(0, 65536, (3, 90, 111), dtype=np.uint16)
    a[1] = a[1] // 257   # narrow range
    a[2, :, :50] = 0     # long runs of equal values (run-length path of the histogram)
    h = ops.histogram16(T(a, dev)).cpu().numpy().view(np.uint32)
    for i in range(3):
        assert np.array_equal(h[i], np.bincount(a[i].ravel(), minlength=65536))
    assert np.array_equal(ops.threshold_otsu(T(a, dev)).cpu().numpy(), [int(o.threshold_otsu(f)) for f in a])
    q = [0, 0.01, 0.5, 5, 50, 99.9, 99.99, 100]
    assert np.array_equal(ops.percentile(T(a, dev), q).numpy(), np.stack([np.percentile(f, q) for f in a]))
    ai = (a.astype(np.int32) - 32768).astype(np.int16)
    assert np.array_equal(ops.threshold_otsu(T(ai, dev)).cpu().numpy(), [int(o.threshold_otsu(f)) for f in ai])
    assert np.array_equal(ops.percentile(T(ai, dev), q).numpy(), np.stack([np.percentile(f, q) for f in ai]))


def test_check_inversion_by_histogram(dev):
    """pylinac/core/image.py:899-926."""
    from pylinac_amd.image import ArrayImage

    rng = np.random.default_rng(2)
    a = (rng.normal(60000, 300, (64, 64))).clip(0, 65535).astype(np.uint16)
    a[20:30, 20:30] = 1000  # mostly-high image with a small dark region
    im = ArrayImage(a.copy())
    p = [np.percentile(a, q) for q in (5, 50, 95)]
    expect = abs(p[1] - p[0]) > abs(p[1] - p[2])
    assert im.check_inversion_by_histogram() == expect
    if expect:
        assert np.array_equal(im.array, o.invert(a))


def test_find_peaks_vs_oracle_random(dev):
    from pylinac_amd import ops

    rng = np.random.default_rng(17)
    n_peaks = 0
    for trial in range(80):
        L = int(rng.integers(3, 6000))
        if trial % 4 == 0:
            x = rng.integers(0, 7, L).astype(float)  # plateaus
        else:
            x = np.abs(rng.normal(size=L).cumsum())
        kws = [dict(), dict(threshold=0.3, peak_separation=0.05),
               dict(threshold=0.5, peak_separation=0.02, peak_sort="peak_heights",
                    required_prominence=0.1 * np.ptp(x), max_number=3),
               dict(search_region=(0.2, 0.8), max_number=2), dict(fwxm_height=0.3, max_number=1),
               dict(threshold=0.2, peak_separation=3, peak_sort="widths", max_number=4)]
        kw = dict(kws[trial % len(kws)])
        if trial % 4 == 0:
            kw.pop("peak_separation", None)  # exact ties + distance: order is implementation-defined
            kw.pop("max_number", None)
        i1, p1 = o.find_peaks(x, **kw)
        i2, p2 = ops.find_peaks_batch(T(x, dev), **kw).to_host(0)
        assert np.array_equal(i1, i2), (trial, kw)
        for k in p1:
            assert np.array_equal(p1[k], p2[k]), (trial, k)
        n_peaks += len(i1)
    assert n_peaks > 1000


def test_field_cax_driven_by_tile_maxima_equals_the_full_pass(dev):
    import next_row_checks as checks

    checks.check_field_cax_tile_maxima(dev, big=True)


@pytest.mark.gpu
def test_bb_sweep_run_table_tiers(dev):
    """the sweep's three run-table sizes (three / two / one workgroup per CU) hand frames on by status 5"""
    import next_row_checks as checks

    assert checks.check_bb_sweep_run_table_tiers(dev) == 4


def test_fwxm_search_short_profiles_corner_cases(dev):
    import next_row_checks as checks

    checks.check_fwxm_short_profiles(dev)


def test_find_peaks_batch_rows_are_independent(dev):
    from pylinac_amd import ops

    rng = np.random.default_rng(23)
    x = np.abs(rng.normal(size=(37, 777)).cumsum(axis=1))
    res = ops.find_peaks_batch(T(x, dev), threshold=0.3, peak_separation=0.05)
    for i in range(x.shape[0]):
        i1, p1 = o.find_peaks(x[i], threshold=0.3, peak_separation=0.05)
        i2, p2 = res.to_host(i)
        assert np.array_equal(i1, i2)
        assert np.array_equal(p1["left_ips"], p2["left_ips"]) and np.array_equal(p1["prominences"], p2["prominences"])


def test_peak_capacity_overflow_is_reported_not_hidden(dev):
    from pylinac_amd import _lib, ops

    x = np.tile([0.0, 1.0], 200)
    res = ops.find_peaks_batch(T(x, dev), cap=5)
    assert int(res.status[0]) == 1 and int(res.count[0]) == 5
    with pytest.raises(_lib.PylinacHipError):
        res.to_host(0)


# ------------------------------------------------------------------------ pipeline + full-size tests
def test_epid_pipeline_vs_oracle_small(dev):
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    n, h, w = 5, 200, 264
    fr = epid_open_field_frames(n, h, w, seed0=42, device=dev, field_mm=35.0)
    res = EpidPipeline(n, h, w, dev).run(fr)
    out, prof, rec = o.epid_pipeline(fr.cpu().numpy())
    assert np.array_equal(res.frames.cpu().numpy(), out)
    assert np.array_equal(res.profile.cpu().numpy(), prof)
    got = res.record().cpu().numpy()
    assert np.array_equal(got[:, :3], rec[:, :3])
    assert np.allclose(got, rec, rtol=1e-12, atol=0, equal_nan=True)


def test_epid_pipeline_widths_off_the_fused_paths(dev):
    """Widths the fused stages do not all cover: 270 (even, not a multiple of 8: matrix-core Gaussian with a ragged strip and a
    quad across the edge, then the three separate median / Otsu / threshold launches) and 135 (odd: two-pass float64
    Gaussian as well) give the oracle's frames, profiles and records like any other width."""
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    for w in (270, 135):
        n, h = 3, 100
        fr = epid_open_field_frames(n, h, w, seed0=77, device=dev, field_mm=20.0)
        res = EpidPipeline(n, h, w, dev).run(fr)
        out, prof, rec = o.epid_pipeline(fr.cpu().numpy())
        assert np.array_equal(res.frames.cpu().numpy(), out), w
        assert np.array_equal(res.profile.cpu().numpy(), prof), w
        got = res.record().cpu().numpy()
        assert np.array_equal(got[:, :3], rec[:, :3]), w
        assert np.allclose(got, rec, rtol=1e-12, atol=0, equal_nan=True), w


def test_full_size_frames_vs_oracle_and_properties(dev):
    """BASELINE config #2 size (1024x1024): 2 frames against the oracle end-to-end, then
    size-independent properties on a 24-frame batch: idempotence of thresholding, permutation
    equivariance over the batch, histogram mass, profile/threshold consistency."""
    from pylinac_amd import ops
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    n, h, w = 24, 1024, 1024
    fr = epid_open_field_frames(n, h, w, seed0=1000, device=dev)
    pipe = EpidPipeline(n, h, w, dev)
    res = pipe.run(fr)
    out = res.frames.clone()
    thr = res.threshold.clone()
    prof = res.profile.clone()
    fw = res.fwxm.clone()
    ref_out, ref_prof, ref_rec = o.epid_pipeline(fr[:2].cpu().numpy())
    assert np.array_equal(out[:2].cpu().numpy(), ref_out)
    assert np.array_equal(prof[:2].cpu().numpy(), ref_prof)
    assert np.array_equal(thr[:2].cpu().numpy(), ref_rec[:, 0])
    assert np.allclose(fw[:2].cpu().numpy(), ref_rec[:, 1:], rtol=1e-12, atol=0)
    # idempotence: thresholding the thresholded frame again changes nothing
    again = ops.threshold(out, thr.to(torch.float64))
    assert torch.equal(again.view(torch.int16), out.view(torch.int16))
    # every output pixel is 0 or >= its frame's threshold
    o32 = out.to(torch.int32)
    assert bool(((o32 == 0) | (o32 >= thr[:, None, None])).all())
    # profile == column mean of the output frame (exact in float64)
    assert torch.equal(prof, o32.to(torch.float64).sum(dim=1) / h)
    # histogram mass
    hist = ops.histogram16(out)
    assert bool((hist.to(torch.int64).sum(dim=1) == h * w).all())
    # permutation equivariance: frames are independent
    perm = torch.randperm(n, device=dev)
    res2 = pipe.run(fr.view(torch.int16)[perm].contiguous().view(torch.uint16))
    assert torch.equal(res2.threshold, thr[perm])
    assert torch.equal(res2.frames.view(torch.int16), out.view(torch.int16)[perm])
    assert torch.equal(res2.fwxm, fw[perm])
    # physics sanity: a ~595 px (20 cm) field centred near the middle
    assert bool(((fw[:, 7] > 560) & (fw[:, 7] < 630)).all()) and bool(((fw[:, 6] - 511.5).abs() < 8).all())


def test_gaussian_linearity_property_full_size(dev):
    """Property at full size without the oracle: the float64 pass is linear BEFORE truncation, so
    G(a+b) - G(a) - G(b) stays within the accumulated truncation error of two passes."""
    from pylinac_amd import ops

    g = torch.Generator(device=dev)
    g.manual_seed(1)
    a = torch.randint(0, 30000, (2, 1024, 1024), generator=g, device=dev, dtype=torch.int32)
    b = torch.randint(0, 30000, (2, 1024, 1024), generator=g, device=dev, dtype=torch.int32)

    def u(x):
        return (x & 0xFFFF).to(torch.int16).view(torch.uint16)

    ga = ops.gaussian_filter(u(a), 5).to(torch.int32)
    gb = ops.gaussian_filter(u(b), 5).to(torch.int32)
    gab = ops.gaussian_filter(u(a + b), 5).to(torch.int32)
    d = gab - ga - gb
    assert int(d.min()) >= -1 and int(d.max()) <= 3
    # median commutes with an order-reversing map: median3(65535 - x) == 65535 - median3(x)
    x = a[0:1] * 2
    m1 = ops.median_filter(u(x), 3).to(torch.int32)
    m2 = ops.median_filter(u(65535 - x), 3).to(torch.int32)
    assert torch.equal(m2, 65535 - m1)


# ------------------------------------------------- components, circle profiles, Sobel, WL field CAX
def test_label_fill_centroid_vs_golden(golden, dev):
    from pylinac_amd import ops

    g = golden("misc")
    for k in ["random40", "random_dense", "blobs", "rings", "empty", "full"]:
        m = g[f"mask.{k}"]
        t = T(m[None], dev)
        for conn, sk in ((4, 1), (8, 2)):
            lab, cnt = ops.label(t, conn)
            ref = g[f"label.{k}.conn{sk}"]
            assert np.array_equal(lab[0].cpu().numpy(), ref), (k, conn)
            assert int(cnt[0]) == int(ref.max())
        assert np.array_equal(ops.fill_holes(t, 4)[0].cpu().numpy(), g[f"fill.{k}"]), k
        if m.any():
            rr, cc = np.nonzero(m)
            cen = ops.binary_centroid(t)[0].cpu().numpy()
            assert np.array_equal(cen, [rr.sum() / len(rr), cc.sum() / len(cc), len(rr)])


def test_label_and_fill_vs_scipy_random_batch(dev):
    from scipy import ndimage

    from pylinac_amd import ops

    rng = np.random.default_rng(77)
    for shape, p in [((5, 64, 80), 0.5), ((2, 257, 300), 0.62), ((3, 31, 1), 0.5), ((1, 1, 40), 0.4)]:
        m = (rng.random(shape) > p).astype(np.uint8)
        t = T(m, dev)
        for conn in (4, 8):
            lab, cnt = ops.label(t, conn)
            for i in range(shape[0]):
                ref, num = o.label_like_skimage(m[i], conn)
                assert np.array_equal(lab[i].cpu().numpy(), ref), (shape, conn, i)
                assert int(cnt[i]) == num
        f4 = ops.fill_holes(t, 4).cpu().numpy()
        f8 = ops.fill_holes(t, 8).cpu().numpy()
        for i in range(shape[0]):
            assert np.array_equal(f4[i], ndimage.binary_fill_holes(m[i]).astype(np.uint8))
            assert np.array_equal(f8[i], ndimage.binary_fill_holes(m[i], np.ones((3, 3))).astype(np.uint8))


def test_circle_profiles_vs_reference_golden(golden, dev):
    from pylinac_amd.profile import CircleProfile, CollapsedCircleProfile, Point

    g = golden("misc")
    img16 = g["circle.img16"]
    imgf = img16.astype(float) / 65535.0
    for i, (cx, cy, r, sa, ccw, sr) in enumerate(g["circle.cases"]):
        c = Point(x=cx, y=cy)
        p = CircleProfile(c, r, img16, start_angle=sa, ccw=bool(ccw), sampling_ratio=sr)
        assert p.values.dtype == img16.dtype and np.array_equal(p.values, g[f"circle.{i}.u16"])
        p = CircleProfile(c, r, imgf, start_angle=sa, ccw=bool(ccw), sampling_ratio=sr)
        assert np.array_equal(p.values, g[f"circle.{i}.f64"])
        p = CollapsedCircleProfile(c, r, imgf, start_angle=sa, ccw=bool(ccw), sampling_ratio=sr, width_ratio=0.1,
                                   num_profiles=20)
        assert np.array_equal(p.values, g[f"collapsed.{i}.f64"])
        p = CollapsedCircleProfile(c, r, img16, sampling_ratio=sr, width_ratio=0.05, num_profiles=5)
        assert np.array_equal(p.values, g[f"collapsed.{i}.u16"])
        # MultiProfile API on top of the sampled ring (Starshot / CTP528 use find_peaks on it)
        i1, v1 = p.find_peaks(threshold=0.3, min_distance=0.05)
        i2, v2 = o.multiprofile_find_peaks(g[f"collapsed.{i}.u16"], 0.3, 0.05)
        assert np.array_equal(i1, i2) and np.array_equal(v1, v2)
    with pytest.raises(ValueError, match="not large enough"):
        CircleProfile(Point(x=150, y=80), 100.0, img16)


def test_circle_profile_out_of_bounds_is_zero(dev):
    """A ring that leaves the image on the low side: out-of-range samples read 0, even when only
    fractionally outside (scipy mode='constant')."""
    from pylinac_amd import ops

    rng = np.random.default_rng(1)
    img = rng.integers(1, 65536, (60, 70), dtype=np.uint16)
    got = ops.circle_profile(T(img[None], dev), 20.2, 15.7, [30.0], np.pi * 30 * 2)[0].cpu().numpy()
    ref = o.circle_profile(img, (20.2, 15.7), 30.0)
    assert np.array_equal(got, ref.astype(float)) and (ref == 0).any() and (ref != 0).any()


def test_sobel_vs_scipy(golden, dev):
    from pylinac_amd import ops

    g = golden("misc")
    x = g["sobel.in"]
    assert np.array_equal(ops.sobel(T(x[None], dev), 1)[0].cpu().numpy(), g["sobel.axis1"])
    assert np.array_equal(ops.sobel(T(x[None], dev), 0)[0].cpu().numpy(), g["sobel.axis0"])
    rng = np.random.default_rng(6)
    for dt in (np.float64, np.float32, np.int16):
        a = (rng.normal(0, 300, (2, 37, 53))).astype(dt)
        for ax in (0, 1):
            ref = np.stack([o.sobel(f, ax) for f in a])
            assert np.array_equal(ops.sobel(T(a, dev), ax).cpu().numpy(), ref), (dt, ax)


def test_wl_field_centroid_vs_reference_golden(golden, dev):
    from pylinac_amd.winston_lutz import field_centroids_batch

    g = golden("misc")
    got = field_centroids_batch(T(g["wl.in"], dev)).cpu().numpy()
    assert np.array_equal(got, g["wl.centroid"])
    # seeded extra frames against the oracle, incl. a field with an interior hole and int16 data
    rng = np.random.default_rng(3)
    fr = g["wl.in"].copy()
    fr[0, 95:105, 115:125] = 0
    ref = np.array([o.wl_field_centroid(f) for f in fr])
    assert np.array_equal(field_centroids_batch(T(fr, dev)).cpu().numpy(), ref)
    with pytest.raises(TypeError):  # int16 ground() wraps in the reference itself: refused, not emulated
        field_centroids_batch(T((fr.astype(np.int32) - 32768).astype(np.int16), dev))


# ------------------------------------------------------------------------- CatPhan localisation (a16)
def test_catphan_stages_vs_skimage_golden(golden, dev):
    from pylinac_amd import ct, ops

    g = golden("catphan")
    sl, mm = g["slices"], float(g["mm_per_pixel"])
    t = T(sl, dev)
    sch = ops.scharr(t)
    assert np.array_equal(sch[0].cpu().numpy(), g["0.scharr"])
    gs = ops.gaussian_filter_mode(sch, 1, "nearest")
    assert np.array_equal(gs[0].cpu().numpy(), g["0.gauss"])
    reg = ct.get_regions_batch(t, mm)
    n = len(sl)
    assert np.array_equal(reg["otsu"].cpu().numpy(), [float(g[f"{i}.otsu"]) for i in range(n)])   # device float Otsu
    assert np.array_equal(reg["bw"].cpu().numpy(), np.stack([g[f"{i}.filled"] for i in range(n)]))
    assert np.array_equal(reg["labels"].cpu().numpy(), np.stack([g[f"{i}.labels"] for i in range(n)]))
    stats = reg["stats"].cpu().numpy()
    for i in range(n):
        p = g[f"{i}.props"]          # area, bbox4, centroid2, filled_area, weighted_centroid2
        k = len(p)
        assert int(reg["num"][i]) == k
        s = stats[i, :k]
        assert np.array_equal(s[:, 0], p[:, 0]) and np.array_equal(s[:, 1:5], p[:, 1:5])
        assert np.array_equal(s[:, 5] / s[:, 0], p[:, 5]) and np.array_equal(s[:, 6] / s[:, 0], p[:, 6])
        assert np.array_equal(s[:, 0], p[:, 7])          # filled_area == area after binary_fill_holes
        assert np.allclose(s[:, 8] / s[:, 7], p[:, 8], rtol=1e-9, atol=0)
        assert np.allclose(s[:, 9] / s[:, 7], p[:, 9], rtol=1e-9, atol=0)
    # intermediate masks: '>' threshold and clear_border
    thr = torch.from_numpy(reg["otsu"].cpu().numpy() * 0.8).to(dev)
    bw = ops.compare(reg["edges"], thr, ">")
    assert np.array_equal(bw.cpu().numpy(), np.stack([g[f"{i}.bw"] for i in range(n)]))
    cl = ops.clear_border(bw, min(int(max(sl.shape[1:]) / 100), 3))
    assert np.array_equal(cl.cpu().numpy(), np.stack([g[f"{i}.cleared"] for i in range(n)]))


def test_catphan_phantom_roi_batch(golden, dev):
    from pylinac_amd import ct

    g = golden("catphan")
    sl, mm = g["slices"], float(g["mm_per_pixel"])
    bad = np.concatenate([sl, np.zeros_like(sl[:1]), np.full_like(sl[:1], -1000)])
    bad[4, 100:110, 100:110] = 500        # a small square: edges exist, but no phantom-sized ROI
    out = ct.phantom_roi_batch(T(bad, dev), mm)
    for i in range(len(sl)):
        best = g[f"{i}.best"]              # label, filled_area, centroid r, c, bbox
        assert out[i, 0] == 0 and out[i, 1] == best[0] and out[i, 2] == best[1]
        assert np.array_equal(out[i, 3:5], best[2:4]) and np.array_equal(out[i, 5:8], best[4:7])
        k, row = o.catphan_phantom_roi(sl[i], mm, float(g["catphan_size"]))
        assert k == out[i, 1]
    assert out[3, 0] == 1                  # blank slice: "No edges were found" in the reference
    assert out[4, 0] in (2, 3)             # reference raises ValueError (no ROI of the expected size)
    with pytest.raises(ValueError):
        o.catphan_phantom_roi(bad[4], mm, float(g["catphan_size"]))


def test_clip_compare_hist_uniform_vs_numpy(dev):
    from pylinac_amd import ops

    rng = np.random.default_rng(9)
    a = rng.normal(0, 600, (3, 50, 70))
    assert np.array_equal(ops.clip(T(a, dev), -1000, 1000).cpu().numpy(), np.clip(a, -1000, 1000))
    ai = a.astype(np.int16)
    assert np.array_equal(ops.clip(T(ai, dev), -1000, 1000).cpu().numpy(), np.clip(ai, -1000, 1000))
    for op, f in ((">", np.greater), (">=", np.greater_equal), ("<", np.less), ("<=", np.less_equal)):
        assert np.array_equal(ops.compare(T(a, dev), 100.0, op).cpu().numpy(), f(a, 100.0).astype(np.uint8))
    b = np.abs(a) ** 1.5
    b[1] = np.round(b[1], 0)              # many values exactly on bin edges
    edges = np.stack([np.linspace(f.min(), f.max(), 257) for f in b])
    got = ops.hist_uniform(T(b, dev), T(edges, dev)).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], np.histogram(b[i], bins=256)[0])
    m = (rng.random((50, 70)) > 0.4).astype(np.uint8)
    edges = np.stack([np.linspace(f[m.astype(bool)].min(), f[m.astype(bool)].max(), 257) for f in b])
    got = ops.hist_uniform(T(b, dev), T(edges, dev), T(m, dev)).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], np.histogram(b[i][m.astype(bool)], bins=256)[0])
    mn, mx = ops.minmax_masked(T(b, dev), T(m, dev))
    assert np.array_equal(mn.cpu().numpy(), [f[m.astype(bool)].min() for f in b])
    assert np.array_equal(mx.cpu().numpy(), [f[m.astype(bool)].max() for f in b])


def test_ctp528_style_peak_valley_mtf(dev):
    """pylinac/ct.py:1511-1580: collapsed circle profile -> filter(0.001,'gaussian') -> ground ->
    per-region find_peaks / find_valleys -> Michelson MTF, device path vs the oracle composition."""
    from pylinac_amd import mtf as pmtf
    from pylinac_amd.profile import CollapsedCircleProfile, Point

    size, cx, cy, r = 400, 200.3, 199.1, 120.0
    y, x = np.mgrid[0:size, 0:size].astype(float)
    ang = np.mod(np.arctan2(y - cy, x - cx), 2 * np.pi)
    img = np.full((size, size), 100.0)
    settings = {}
    for k, (a0, a1, npk, lp) in enumerate([(0.05, 0.17, 2, 0.1), (0.2, 0.32, 3, 0.2), (0.35, 0.47, 4, 0.3), (0.5, 0.62, 5, 0.4)]):
        seg = (ang >= a0 * 2 * np.pi) & (ang < a1 * 2 * np.pi)
        phase = (ang - a0 * 2 * np.pi) / ((a1 - a0) * 2 * np.pi)
        img[seg] = 100 + (900 - 150 * k) * (0.5 + 0.5 * np.cos(2 * np.pi * (npk + 0.0) * phase[seg] - np.pi))
        settings[f"region {k+1}"] = {"start": a0 - 0.01, "end": a1 + 0.01, "num peaks": npk, "num valleys": npk - 1,
                                     "peak spacing": 0.01 + 0.002 * (4 - npk), "lp/mm": lp}
    img += np.random.default_rng(0).normal(0, 2, img.shape)
    img = np.round(img).astype(np.int16)
    kw = dict(start_angle=0.0, ccw=False, sampling_ratio=2, width_ratio=0.04, num_profiles=20)
    prof = CollapsedCircleProfile(Point(x=cx, y=cy), r, img, **kw)
    prof.filter(0.001, kind="gaussian")
    prof.ground()
    got = pmtf.peak_valley_mtf(prof, settings)
    # oracle composition
    vals = o.collapsed_circle_profile(img, (cx, cy), r, **kw)
    vals = o.ground(o.filter(vals, 0.001, "gaussian"))
    assert np.array_equal(np.asarray(prof.values), vals)
    maxs, mins = [], []
    for v in settings.values():
        i, pv = o.multiprofile_find_peaks(vals, min_distance=v["peak spacing"], max_number=v["num peaks"],
                                          search_region=(v["start"], v["end"]))
        assert len(pv) == v["num peaks"]
        maxs.append(pv.mean())
        _, vv = o.multiprofile_find_valleys(vals, min_distance=v["peak spacing"], max_number=v["num valleys"],
                                            search_region=(min(i), max(i)))
        mins.append(vv.mean())
    assert got.maximums == maxs and got.minimums == mins
    # rMTF restated here from the reference's formulas (pylinac/core/mtf.py:62-72 Michelson contrast per region, normalised to
    # the first region; :91-96 relative_resolution = scipy interp1d of spacing over rMTF) -- not through pylinac_amd.mtf
    from scipy.interpolate import interp1d

    lps = [s["lp/mm"] for s in settings.values()]
    mtfs = {lp: (np.nanmax(np.array((mx, mn))) - np.nanmin(np.array((mx, mn)))) / (np.nanmax(np.array((mx, mn))) + np.nanmin(np.array((mx, mn))))
            for lp, mx, mn in zip(lps, maxs, mins)}
    norm = {k: v / mtfs[lps[0]] for k, v in sorted(mtfs.items())}
    assert got.norm_mtfs == norm
    want50 = float(interp1d(list(norm.values()), list(norm.keys()), fill_value="extrapolate")(0.5))
    assert got.relative_resolution(50) == want50


# --------------------------------------------------------------- picket fence (BASELINE config #3)
def test_picket_fence_batch_vs_reference_analyze(golden, dev):
    """Device pipeline against the reference's real PicketFence.analyze() output (golden) and against
    the oracle for the windows the reference later drops."""
    from pylinac_amd import picketfence as ppf

    g = golden("picketfence")
    for k in (0, 1):
        raw, dpmm = g[f"{k}.cropped"], float(g[f"{k}.dpmm"])
        res = ppf.analyze_batch(T(raw[None], dev), dpmm)
        ref = o.pf_measure(o.normalize(o.ground(raw)), dpmm)
        P = len(ref["peak_idxs"])
        assert int(res.picket_count[0]) == P
        assert np.array_equal(res.picket_idx[0, :P].cpu().numpy(), ref["peak_idxs"])
        assert float(res.spacing[0]) == ref["spacing"] == float(g[f"{k}.spacing"])
        assert res.leaf_nums == [n for n, _, _ in ref["leaves"]]
        pos = res.position[0, :, :P].cpu().numpy()
        assert np.array_equal(np.isnan(pos), np.isnan(ref["position"]))
        assert np.array_equal(pos[~np.isnan(pos)], ref["position"][~np.isnan(pos)])
        idx = {n: i for i, n in enumerate(res.leaf_nums)}
        for leaf, picket, p, _ in g[f"{k}.meas"]:          # what the reference itself measured
            assert pos[idx[int(leaf)], int(picket)] == p
        st = res.status[0].cpu().numpy()
        assert (st[:, P:] == 1).all() and set(np.unique(st[:, :P])) <= {0, 2}


def test_picket_fence_left_right_and_separate_leaves(golden, dev):
    import next_row_checks as checks

    checks.check_pf_orientation_device(golden("picketfence_orient"), dev)


def test_picket_fence_batch_of_frames(dev):
    from pylinac_amd import picketfence as ppf
    from tests.golden.make_golden import pf_frame

    frames = np.stack([pf_frame(300, 520, 0.78125, 2100 + i)[4:-4, 4:-4] for i in range(5)])
    res = ppf.analyze_batch(T(np.ascontiguousarray(frames), dev), 1 / 0.78125)
    for i, f in enumerate(frames):
        ref = o.pf_measure(o.normalize(o.ground(f)), 1 / 0.78125)
        P = len(ref["peak_idxs"])
        pos = res.position[i, :, :P].cpu().numpy()
        assert np.array_equal(np.isnan(pos), np.isnan(ref["position"]))
        assert np.array_equal(pos[~np.isnan(pos)], ref["position"][~np.isnan(pos)])


# ------------------------------------------------------------------------------- BB finder (a13)
def test_find_features_batch_vs_reference_golden(golden, dev):
    """Device sweep against the reference's own find_features (scikit-image 0.18.3, py3.9 helper):
    same first level, same number of features, weighted centroids within 1e-12 relative
    (north_star asks 1e-5)."""
    from pylinac_amd import features as pf

    g = golden("features")
    dpmm = float(g["dpmm"])
    wins = np.stack([o.invert(g[f"{i}.window"]) for i in range(6)])
    # windows 4 / 5 hold two BBs (near: suppressed as a same-level duplicate; far: both reported)
    for sel in ([0, 1, 2, 3], [4], [5]):
        maxn, minsep = int(g["maxn"][sel[0]]), float(g["minsep"][sel[0]])
        res = pf.find_features_batch(T(wins[sel], dev), dpmm, 2.5, 0.5, max_number=maxn, min_separation_mm=minsep)
        assert int(res["status"].abs().sum()) == 0
        for j, i in enumerate(sel):
            ref_pts, ref_level = o.find_features_restated(wins[i], dpmm, 2.5, 0.5, max_number=maxn,
                                                          min_separation_mm=minsep)
            assert int(res["count"][j]) == len(g[f"{i}.points"]) == len(ref_pts)
            assert int(res["level"][j]) == ref_level
            got = res["xy"][j, : len(ref_pts)].cpu().numpy()
            assert np.allclose(got, g[f"{i}.points"], rtol=1e-12, atol=0), i
    # nothing BB-like: the reference raises ValueError("Couldn't find the minimum number of disks")
    rng = np.random.default_rng(0)
    flat = 0.5 + rng.normal(0, 0.01, (1, 134, 134))
    r2 = pf.find_features_batch(T(flat, dev), dpmm, 2.5, 0.5)
    assert int(r2["count"][0]) == 0 and int(r2["level"][0]) == -1
    # the one-launch sweep and the level-by-level path agree bit for bit
    a = pf.find_features_batch(T(wins[:4], dev), dpmm, 2.5, 0.5)
    b = pf.find_features_batch(T(wins[:4], dev), dpmm, 2.5, 0.5, level_by_level=True)
    for key in ("xy", "count", "level", "status"):
        assert torch.equal(a[key], b[key]), key
    with pytest.raises(ValueError):
        o.find_features_restated(flat[0], dpmm, 2.5, 0.5)


def test_ctp528_batch_vs_reference_golden(golden, dev):
    """config #5: the per-slice spatial-resolution record against the reference's own CTP528CP504 on a synthetic volume."""
    import next_row_checks as checks

    checks.check_ctp528_batch(golden, dev)


def test_field_cax_fused_vs_scipy(dev):
    """a14: pl_field_cax (threshold -> fill holes -> centre of mass without a mask or label plane) against scipy."""
    import next_row_checks as checks

    checks.check_field_cax(dev)


def test_wl_analyze_batch_vs_reference_golden(golden, dev):
    """config #4 / north_star n1: the composed per-image Winston-Lutz path against the reference's own sequence."""
    import next_row_checks as checks

    checks.check_wl_analyze_batch(golden, dev)


def test_wl_bb_centroids_batch_vs_oracle(dev):
    """WLBaseImage.find_bb_centroids (pylinac/winston_lutz.py:788-806) on synthetic WL frames: window
    crop (floor/ceil), frame-level ground/normalize, invert, sweep -- against the oracle composition."""
    import math

    from scipy import ndimage as ndi

    from pylinac_amd import features as pf

    rng = np.random.default_rng(12)
    dpmm, n, size = 1 / 0.336, 3, 400
    frames = []
    for i in range(n):
        y, x = np.mgrid[0:size, 0:size].astype(float)
        cy, cx = size / 2 + rng.uniform(-5, 5), size / 2 + rng.uniform(-5, 5)
        img = np.full((size, size), 1500.0)
        img[int(cy - 30):int(cy + 30), int(cx - 30):int(cx + 30)] = 42000.0
        img[np.hypot(y - (cy + rng.uniform(-4, 4)), x - (cx + rng.uniform(-4, 4))) < 2.5 * dpmm] *= 0.3
        img = ndi.gaussian_filter(img, 1.5) + rng.normal(0, 120, img.shape)
        frames.append(np.clip(np.round(img), 0, 65535))
    frames = np.stack(frames).astype(np.uint16)
    res = pf.bb_centroids_batch(T(frames, dev), dpmm, 5.0)
    tol = float(np.interp(5.0, (1.5, 30), (2, 4)))
    for i in range(n):
        arr = o.normalize(o.ground(frames[i]))
        win = (40 + 5.0) * dpmm
        left = max(math.floor(size / 2 - win / 2), 0)
        right = math.ceil(size / 2 + win / 2)
        top, bottom = left, right
        sample = o.invert(arr[top:bottom, left:right])
        pts, level = o.find_features_restated(sample, dpmm, 2.5, tol)
        assert int(res["count"][i]) == len(pts) == 1 and int(res["level"][i]) == level
        got = res["xy"][i, 0].cpu().numpy()
        assert np.allclose(got, [pts[0][0] + left, pts[0][1] + top], rtol=1e-12, atol=0)


def _gaussian_cases(dev):
    """(frames, sigmas) covering sparse undecided pixels, list overflow, zero shortcut, ragged shapes."""
    from pylinac_amd.synthetic import epid_open_field_frames

    cases = [(epid_open_field_frames(3, 1024, 1024, seed0=77, device=dev).cpu().numpy(), (1, 2, 3, 5))]
    rng = np.random.default_rng(5)
    for shape in [(2, 37, 530), (1, 9, 1026), (3, 3, 6), (1, 301, 77), (2, 64, 64), (1, 1, 40), (1, 40, 2)]:
        smooth = 30000 + 8000 * np.sin(np.arange(shape[2]) / 17.0)[None, None, :] * np.cos(
            np.arange(shape[1]) / 11.0)[None, :, None]
        b = np.clip(smooth + rng.normal(0, 300, shape), 0, 65535).astype(np.uint16)
        b[0, : shape[1] // 3, : shape[2] // 2] = 0
        b[-1, shape[1] // 2:, shape[2] // 2:] = 41234
        cases.append((b, (1, 2, 3, 5)))
        cases.append(((b.astype(np.int32) - 31000).astype(np.int16), (2, 5)))
    full = rng.integers(0, 65536, (2, 90, 1100), dtype=np.uint16)   # full-range noise: list overflow
    cases.append((full, (2, 5)))
    cases.append((np.full((1, 70, 260), 65535, dtype=np.uint16), (1, 5)))
    return cases


def _check_gaussian_cases(dev):
    from pylinac_amd import ops

    for arr, sigmas in _gaussian_cases(dev):
        for sigma in sigmas:
            ref = np.stack([o.filter(f, sigma, "gaussian") for f in arr])
            got = ops.gaussian_filter(T(arr, dev), sigma).cpu().numpy()
            assert np.array_equal(got, ref), (arr.shape, arr.dtype, sigma)


def test_gaussian_default_kernels_on_epid_and_ragged_frames(dev):
    """Default dispatch (register-window packed-float32 decision kernels where width / alignment / radius allow, the
    float64 kernels elsewhere -- the ragged shapes here take those): EPID-like frames, zero regions, constant blocks,
    full-range noise (fix-list overflow -> whole-tile recompute), ragged shapes, int16 plateaus."""
    _check_gaussian_cases(dev)


def test_median_consumed_on_the_fly_vs_scipy_and_oracle(dev):
    """pl_median3_otsu16 / pl_median3_threshold_colsum_u16 (the EPID pipeline's stages after the Gaussian: the 3x3 median is
    computed inside the Otsu histogram kernel and again inside the threshold + column-sum kernel, the median plane is never
    written) against scipy's median_filter + the oracle's Otsu + numpy: EPID-like frames, a frame with full-range noise (does
    not fit the one-pass window: flagged, taken by the full-range kernel), a constant frame, widths that leave the last
    wave partly idle, a height that is not a multiple of 16, int16."""
    from scipy import ndimage

    from pylinac_amd import ops

    rng = np.random.default_rng(77)
    # (3, 512, 1024): a batch this small runs EIGHT workgroups per frame in the window kernel (bands of rows, merged table)
    for shape in ((3, 200, 264), (2, 37, 1040), (4, 130, 64), (1, 16, 8), (3, 512, 1024)):
        for dt in (np.uint16, np.int16):
            a = (rng.integers(2000, 2600, shape) + (np.arange(shape[2]) > shape[2] // 2) * 9000).astype(np.int64)
            a[-1] = rng.integers(0, 65536, shape[1:])                # full range: the packed-counter kernel
            if shape[0] > 2:
                a[1] = 12345                                          # constant frame
            a = (a - (32768 if dt == np.int16 else 0)).astype(dt)
            t = torch.from_numpy(a).to(dev)
            med = np.stack([ndimage.median_filter(f, size=3) for f in a])
            thr, mn, mx, flag = ops.median3_otsu16(t)
            assert np.array_equal(thr.cpu().numpy(), np.array([o.threshold_otsu(f) for f in med])), (shape, dt)
            assert np.array_equal(mn.cpu().numpy(), med.reshape(shape[0], -1).min(1)), (shape, dt)
            assert np.array_equal(mx.cpu().numpy(), med.reshape(shape[0], -1).max(1)), (shape, dt)
            assert int(flag[-1]) == 1 or shape[1] * shape[2] < 1000, (shape, dt, flag.cpu().tolist())
            # the frames that fit the window must be finished by the window kernel itself (a silent trip through the
            # full-range kernel gives the same numbers 50 % slower: round 4 lost a stage that way to a stray static LDS byte)
            assert not flag[:-1].cpu().numpy().any(), (shape, dt, flag.cpu().tolist())
            if dt == np.uint16:
                cut = torch.from_numpy(np.array([int(np.percentile(f, 40)) for f in med], dtype=np.int32)).to(dev)
                out, cs = ops.median3_threshold_colsum_u16(t, cut)
                want = np.where(med.astype(np.int64) >= cut.cpu().numpy()[:, None, None], med, 0).astype(np.uint16)
                assert np.array_equal(out.cpu().numpy(), want), shape
                assert np.array_equal(cs.cpu().numpy(), want.astype(np.int64).sum(1)), shape


def test_full_range_otsu_counter_overflow_1024(dev):
    """The full-range Otsu kernel (65 536 packed 16-bit counters, guard bit, 32 768-count folds) under REAL concurrency: 1024 x
    1024 frames whose histogram has bins far beyond 32 767 -- one value on 75 % of the pixels scattered among full-range noise
    (single adds cross the guard ~24 times for one key while 1 024 threads race), flat halves (wave-uniform bulk adds), two
    values at the ends of the range, an EPID frame stretched to the full range -- against the oracle's Otsu on scipy's median;
    plain frames (pl_otsu16) too.  uint16 and int16."""
    from scipy import ndimage

    from pylinac_amd import ops
    from pylinac_amd.synthetic import epid_open_field_frames

    rng = np.random.default_rng(19)
    h = w = 1024
    epid = epid_open_field_frames(1, h, w, seed0=7100).numpy()[0].astype(np.float64)
    lo, hi = np.quantile(epid[::16], [0.01, 0.99])
    stretched = np.clip(np.round((epid - lo) * (64500.0 / (hi - lo)) + 500.0), 0, 65535)
    for dt in (np.uint16, np.int16):
        off = 32768 if dt == np.int16 else 0
        a = rng.integers(0, 65536, (4, h, w))
        a[0][rng.random((h, w)) < 0.75] = 777
        a[1][:, : w // 2] = 40000
        a[1][rng.random((h, w)) < 0.3] = 40001
        a[2] = np.where(rng.random((h, w)) < 0.5, 12, 65535)
        a[3] = stretched
        a = (a - off).astype(dt)
        t = torch.from_numpy(a).to(dev)
        thr, mn, mx = ops.otsu16(t)
        assert np.array_equal(thr.cpu().numpy(), np.array([o.threshold_otsu(f) for f in a])), dt
        assert np.array_equal(mn.cpu().numpy(), a.reshape(4, -1).min(1)) and np.array_equal(mx.cpu().numpy(), a.reshape(4, -1).max(1))
        med = np.stack([ndimage.median_filter(f, size=3) for f in a])
        thr, mn, mx, flag = ops.median3_otsu16(t)
        assert flag.cpu().numpy().all()
        assert np.array_equal(thr.cpu().numpy(), np.array([o.threshold_otsu(f) for f in med])), dt
        assert np.array_equal(mn.cpu().numpy(), med.reshape(4, -1).min(1)) and np.array_equal(mx.cpu().numpy(), med.reshape(4, -1).max(1))


def test_pipeline_fused_stages_equal_separate_ops_at_baseline_size(dev):
    """BASELINE configs[1] size (256 frames of 1024 x 1024): the pipeline's fused stages (both Gaussian axes in one launch;
    3x3 median consumed on the fly by the Otsu histogram and by the threshold + column sums) against the composition of
    the separate entry points (pl_gaussian1d twice, pl_median2d, pl_otsu16, pl_threshold_colsum_u16) -- different kernels
    for every stage, 268 M pixels, bit for bit."""
    from pylinac_amd import _lib, ops
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    n, h, w = 256, 1024, 1024
    fr = epid_open_field_frames(n, h, w, seed0=7000, device=dev)
    res = EpidPipeline(n, h, w, dev).run(fr)
    wts, hw, rad = ops._device_weights(5, dev)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    tmp, g = torch.empty_like(fr), torch.empty_like(fr)
    assert lib.pl_gaussian1d(fr.data_ptr(), tmp.data_ptr(), _lib.PL_U16, n, h, w, 0, wts.data_ptr(), hw.ctypes.data, rad, st) == 0
    assert lib.pl_gaussian1d(tmp.data_ptr(), g.data_ptr(), _lib.PL_U16, n, h, w, 1, wts.data_ptr(), hw.ctypes.data, rad, st) == 0
    med = ops.median_filter(g, 3)
    thr, _, _ = ops.otsu16(med)
    out, cs = ops.threshold_colsum_u16(med, thr)
    assert torch.equal(res.threshold, thr)
    assert torch.equal(res.frames.view(torch.int16), out.view(torch.int16))
    assert torch.equal(res.profile, cs.to(torch.float64) / h)


def test_gaussian_marching_strip_kernel_vs_scipy(dev):
    """gauss2d_mm (both axes in one launch, exact integer arithmetic on the matrix cores): frames of at least 64 x 64 with an
    even width take it.  Several strips with a partial last one (also ragged: width % 16 != 0, and with a column quad across
    the right edge: width % 4 == 2, e.g. the aS1200's 1190), several row segments, mirrored border quads, EPID-like
    content, zero / constant / saturated blocks (whole-tile constant path), full-range noise, int16, every sigma the
    analyzers use plus radius 24; 1024 x 1024 frames against scipy itself."""
    from scipy import ndimage

    from pylinac_amd import ops

    rng = np.random.default_rng(2024)
    cases = []
    for shape in ((2, 300, 528), (1, 64, 64), (3, 130, 1040), (1, 1024, 1024), (2, 150, 1190), (2, 70, 66), (1, 200, 300)):
        smooth = ndimage.gaussian_filter(rng.integers(0, 65535, shape).astype(float), (0, 6, 6))
        a = np.clip(20000 + 6.0 * (smooth - 32767) + rng.normal(0, 250, shape), 0, 65535).astype(np.uint16)
        a[0, : shape[1] // 3, : shape[2] // 2] = 0
        a[-1, shape[1] // 2:, shape[2] // 2:] = 41234
        a[0, shape[1] // 2:, : shape[2] // 4] = 65535
        cases.append(a)
    cases.append(rng.integers(0, 65536, (2, 200, 272), dtype=np.uint16))
    cases.append((cases[0].astype(np.int32) - 31000).astype(np.int16))
    cases.append(np.full((1, 96, 256), 65535, dtype=np.uint16))
    for a in cases:
        for sigma in ((5,) if a.shape[1] >= 1024 else (1, 2, 3, 5, 6)):
            ref = np.stack([ndimage.gaussian_filter(f, sigma) for f in a])
            got = ops.gaussian_filter(T(a, dev), sigma).cpu().numpy()
            assert np.array_equal(got, ref), (a.shape, a.dtype, sigma, int((got != ref).sum()))


def test_gaussian_marching_strip_equals_two_pass_kernels_full_size(dev):
    """At BASELINE size (64 frames of 1024 x 1024) without the oracle: the one-launch kernel (pl_gaussian2d) and the two
    single-axis launches (pl_gaussian1d twice: a different kernel family with a different decision rule) agree bit for
    bit on EPID frames -- 67 M outputs, where a decision bound that is too tight by one part in a million would show."""
    from pylinac_amd import _lib, ops
    from pylinac_amd.synthetic import epid_open_field_frames

    fr = epid_open_field_frames(64, 1024, 1024, device=dev)
    one = ops.gaussian_filter(fr, 5)
    wts, hw, rad = ops._device_weights(5, dev)
    lib = _lib.load()
    tmp, two = torch.empty_like(fr), torch.empty_like(fr)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.pl_gaussian1d(fr.data_ptr(), tmp.data_ptr(), _lib.PL_U16, 64, 1024, 1024, 0, wts.data_ptr(), hw.ctypes.data, rad, st) == 0
    assert lib.pl_gaussian1d(tmp.data_ptr(), two.data_ptr(), _lib.PL_U16, 64, 1024, 1024, 1, wts.data_ptr(), hw.ctypes.data, rad, st) == 0
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))


def test_gaussian_without_host_taps_runs_the_float64_kernels(dev):
    """pl_gaussian2d with h_weights = NULL: no device-to-host fetch, no stream synchronisation -- the float64
    kernels read the device copy of the taps and produce the same frames."""
    from pylinac_amd import _lib, ops

    rng = np.random.default_rng(77)
    a = rng.integers(0, 65536, (2, 96, 160), dtype=np.uint16)
    t = T(a, dev)
    out, tmp = torch.empty_like(t), torch.empty_like(t)
    wts, _, rad = ops._device_weights(3, dev)
    rc = _lib.load().pl_gaussian2d(t.data_ptr(), out.data_ptr(), tmp.data_ptr(), _lib.PL_U16, 2, 96, 160,
                                   wts.data_ptr(), None, rad, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert np.array_equal(out.cpu().numpy(), np.stack([o.filter(f, 3, "gaussian") for f in a]))


# ------------------------------------------------------------------------ spectral measures (a18)
def test_nps_and_radial_average_vs_reference_golden(golden, dev):
    """pl_nps2d / pl_radial_average against pylinac.core.nps (golden from the reference itself): the 2-D
    spectrum within 1e-9 of its maximum (plain DFT vs pocketfft summation order), the radial average of a
    GIVEN spectrum bit-identical (same bins, numpy's accumulation order), scalars to 1e-9."""
    from pylinac_amd import nps

    g = golden("spectral")
    rois = {"single": (1, ["roi1"]), "two": (0.5, ["roi1", "roi2"]), "ragged": (0.39, ["roi2", "roi3"]),
            "hu": (0.48, None)}
    for k, (px, names) in rois.items():
        rr = [g[n] for n in names] if names else list(g["hu"])
        got = nps.noise_power_spectrum_2d(px, rr, device=dev).cpu().numpy()
        ref = g[f"nps_{k}"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-9 * ref.max(), k
        # radial average of the reference's spectrum: bit-identical bins
        one = nps.noise_power_spectrum_1d(torch.from_numpy(ref).to(dev))
        assert np.array_equal(one.cpu().numpy(), g[f"nps1d_{k}"]), k
        # and end to end from the device spectrum
        one2 = nps.noise_power_spectrum_1d(torch.from_numpy(got).to(dev))
        assert np.allclose(one2.cpu().numpy(), g[f"nps1d_{k}"], rtol=0, atol=1e-9 * g[f"nps1d_{k}"].max())
        assert abs(nps.average_power(one2) - g[f"scalars_{k}"][0]) < 1e-9
        assert nps.max_frequency(one2) == g[f"scalars_{k}"][1]
    odd = nps.noise_power_spectrum_2d(1, [g["roi1"][:-1, :-1]], device=dev).cpu().numpy()
    assert np.abs(odd - g["nps_odd"]).max() <= 1e-9 * g["nps_odd"].max()
    stacked = nps.noise_power_spectrum_2d(0.48, torch.from_numpy(g["hu"]).to(dev)).cpu().numpy()
    assert np.abs(stacked - g["nps_hu"]).max() <= 1e-9 * g["nps_hu"].max()
    assert np.array_equal(nps.radial_average(g["rect"], device=dev).cpu().numpy(), g["radial_rect"])
    assert np.array_equal(nps.radial_average(np.ones((300, 300)), device=dev).cpu().numpy(), g["radial_ones"])
    with pytest.raises(ValueError):
        nps.noise_power_spectrum_1d(torch.zeros(5, device=dev))


def test_esf_mtf_vs_reference_golden(golden, dev):
    """EdgeSpreadFunctionMTF (pl_esf_mtf) on the reference's known-answer inputs
    (tests_basic/core/test_mtf.py:59-132: ideal steps of 8 / 6 / 256 samples, Hann / Kaiser / Tukey / no
    window, sample spacing, every padding mode) and on blurred noisy edges: frequency axis exact, MTF within
    1e-12 absolute of the reference, resolutions to 1e-9; same ValueErrors."""
    from scipy.signal import windows

    from pylinac_amd import mtf

    g = golden("spectral")
    kws = {"single": {}, "multi": {}, "spacing": dict(sample_spacing=10),
           "kaiser": dict(windowing=windows.kaiser, beta=0.5), "shift_none": dict(windowing=None),
           "shift_hann": {}, "shift_tukey": dict(windowing=windows.tukey, alpha=0.2),
           "pad_none": dict(padding_mode="none"), "pad_fixed": dict(padding_mode="fixed", num_samples=100),
           "blur": dict(sample_spacing=0.25)}
    for name, kw in kws.items():
        esf = [g[f"esf_{name}.in{i}"] for i in range(sum(k.startswith(f"esf_{name}.in") for k in g.files))]
        m = mtf.EdgeSpreadFunctionMTF(esf, device=dev, **kw)
        assert np.array_equal(m.freq, g[f"esf_{name}.freq"]), name
        assert np.abs(m.mtf - g[f"esf_{name}.mtf"]).max() < 1e-12, name
        assert np.abs(np.array(m._mtf) - g[f"esf_{name}.each"]).max() < 1e-12, name
        res = [m.relative_resolution(t) for t in (30, 50, 80)]
        assert np.allclose(res, g[f"esf_{name}.res"], rtol=1e-9, atol=1e-12), name
    ideal = mtf.EdgeSpreadFunctionMTF([np.append(np.zeros(4), np.ones(4))], device=dev)
    assert np.allclose(ideal.mtf, np.cos(np.pi * ideal.freq))
    with pytest.raises(ValueError):
        mtf.EdgeSpreadFunctionMTF([np.zeros(8), np.zeros(6)], padding_mode="none", device=dev)
    with pytest.raises(ValueError):
        mtf.EdgeSpreadFunctionMTF([np.zeros(8)], padding_mode="fixed", num_samples=4, device=dev)
    with pytest.raises(ValueError):
        ideal.relative_resolution(101)


# ------------------------------------------------------------------------------ SingleProfile (a11)
def test_single_profile_vs_reference_frozen_fixtures(golden, dev):
    """Device SingleProfile (pl_interp1d + ground/normalize kernels + pl_find_peaks) on the reference's 20 frozen
    detector profiles x 6 resampling modes and on EPID-style profiles with dpmm / normalisation / centering
    options: resampled values bit-identical for NONE / LINEAR, 1e-10 for SPLINE (not-a-knot spline computed
    directly, scipy evaluates the B-spline form); every scalar of fwxm_data / field_data to 1e-9; the six
    protocol metrics to 1e-9 of the reference run AND of the reference's frozen exports
    (tests_basic/core/test_profile.py:2546-2688); frozen field geometry to 1e-4."""
    from pylinac_amd import profile as pp
    from tests.test_oracle_golden import _SP_EPID, _SP_MODES, _sp_calculators, _sp_check

    g = golden("single_profile")
    calcs = _sp_calculators()
    for i in range(20):
        for mode, (interp, use_x) in _SP_MODES.items():
            p = pp.SingleProfile(g[f"fx{i}.y"], x_values=g[f"fx{i}.x"] if use_x else None, interpolation=interp)
            vtol = 1e-10 if interp == "Spline" else 0
            got, fd = _sp_check(g, f"fx{i}.{mode}", p, calcs, vtol=vtol, ftol=1e-9)
            assert np.allclose(got, g[f"fx{i}.{mode}.frozen_metrics"], rtol=0, atol=1e-9), (i, mode)
            if mode == "none":
                for k, v in zip(g[f"fx{i}.frozen_field_keys"], g[f"fx{i}.frozen_field"]):
                    assert abs(float(fd[str(k)]) - v) < 1e-4, (i, k)
    for name, kw in _SP_EPID.items():
        vtol = 1e-10 if kw.get("interpolation") == "Spline" else 0
        p = pp.SingleProfile(g["epid.y"].copy(), **kw)
        _sp_check(g, f"epid.{name}", p, calcs, vtol=vtol, ftol=1e-9)
        if "dpmm" in kw:
            fd = p.field_data()
            assert abs(fd["width (exact) mm"] - fd["width (exact)"] / kw["dpmm"]) < 1e-12
    # error behaviour
    with pytest.raises(ValueError):
        pp.SingleProfile(np.arange(10.0), x_values=np.arange(10.0)[::-1])
    with pytest.raises(ValueError):
        pp.SingleProfile(g["epid.y"]).field_data(in_field_ratio=0.2, slope_exclusion_ratio=0.5)
    with pytest.raises(ValueError):
        pp.SingleProfile(g["epid.y"]).penumbra(lower=80, upper=20)


def test_interp1d_batch_vs_scipy(dev):
    """pl_interp1d on a batch: shared and per-profile abscissae (non-uniform), extrapolation on both sides,
    queries exactly on nodes; linear bit-identical to scipy's interp1d, cubic within 1e-10 relative."""
    from scipy.interpolate import interp1d

    from pylinac_amd import ops

    rng = np.random.default_rng(4)
    for length in (4, 5, 33, 257):
        x = np.cumsum(rng.uniform(0.2, 1.7, (3, length)), axis=1)
        y = 100 * np.exp(-((x - x.mean(1, keepdims=True)) / (0.2 * np.ptp(x, axis=1, keepdims=True))) ** 4) + \
            rng.normal(0, 0.5, (3, length))
        xq = np.sort(np.concatenate([np.linspace(x.min() - 0.4, x.max() + 0.4, 7 * length), x[0, ::3]]))
        for kind in ("linear", "cubic"):
            got = ops.interp1d(T(x, dev), T(y, dev), T(xq, dev), kind=kind).cpu().numpy()
            ref = np.stack([interp1d(x[i], y[i], kind=kind, bounds_error=False, fill_value="extrapolate")(xq)
                            for i in range(3)])
            if kind == "linear":
                assert np.array_equal(got, ref), length
            else:
                assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), length
            shared = ops.interp1d(T(x[0], dev), T(y, dev), T(xq, dev), kind=kind).cpu().numpy()
            ref0 = np.stack([interp1d(x[0], y[i], kind=kind, bounds_error=False, fill_value="extrapolate")(xq)
                             for i in range(3)])
            assert np.abs(shared - ref0).max() <= 1e-10 * np.abs(ref0).max()
    with pytest.raises(Exception):
        ops.interp1d(T(np.arange(3.0), dev), T(np.arange(3.0), dev), T(np.arange(3.0), dev), kind="cubic")


def test_find_fields_batch_vs_reference_golden(golden, dev):
    """Device GlobalSizedFieldLocator (pl_fields_level) against the reference's own run under scikit-image
    0.18.3: same fields, same order, centroids to 1e-12; same first level as the oracle; a frame without any
    matching field reports count 0 (the reference's ValueError)."""
    from pylinac_amd import features as pf

    g = golden("fields")
    dpmm = float(g["dpmm"])
    for sel in ([0, 1], [2]):
        frames = np.stack([g[f"{i}.frame"].astype(np.float64) for i in sel])
        i0 = sel[0]
        res = pf.find_fields_batch(T(frames, dev), dpmm, float(g["fw"][i0]), float(g["fh"][i0]), float(g["tol"][i0]),
                                   max_number=int(g["maxn"][i0]))
        assert int(res["status"].abs().sum()) == 0
        for j, i in enumerate(sel):
            ref = g[f"{i}.points"]
            pts, lvl = o.find_fields_restated(frames[j], dpmm, g["fw"][i], g["fh"][i], g["tol"][i],
                                              max_number=int(g["maxn"][i]))
            assert int(res["count"][j]) == len(ref) == len(pts)
            assert int(res["level"][j]) == lvl
            assert np.allclose(res["xy"][j, : len(ref)].cpu().numpy(), ref, rtol=1e-12, atol=0), i
    # pixel-unit construction (is_from_physical=False divides by dpmm, image.py:829-832)
    f2 = g["2.frame"].astype(np.float64)[None]
    r2 = pf.find_fields_batch(T(f2, dev), dpmm, 15.0 * dpmm, 15.0 * dpmm, 1.5 * dpmm, max_number=1,
                              is_from_physical=False)
    assert np.allclose(r2["xy"][0, 0].cpu().numpy(), g["2.points"][0], rtol=1e-12)
    none = pf.find_fields_batch(T(f2, dev), dpmm, 40.0, 40.0, 1.0, max_number=1)
    assert int(none["count"][0]) == 0 and int(none["level"][0]) == -1
    with pytest.raises(ValueError):
        o.find_fields_restated(f2[0], dpmm, 40.0, 40.0, 1.0, max_number=1)


# -------------------------------------------------------------------------------- ROI statistics (f3)
def test_disk_roi_stats_vs_reference_golden(golden, dev):
    """pl_roi_stats against the reference's own DiskROI (scikit-image 0.18.3 draw.disk): pixel count, min, max and
    median exact, mean / std to 1e-12; batch form with shared and per-frame ROIs; the DiskROI class incl.
    from_phantom_center; rectangle windows against numpy slices; ROIs leaving the frame are reported."""
    from pylinac_amd import roi

    g = golden("roi")
    rois = g["rois"]
    for name, arr in (("i16", g["slice_i16"]), ("f64", g["slice_f32"].astype(np.float64))):
        ref = g[f"stats_{name}"]
        out, status = roi.disk_roi_stats_batch(T(np.stack([arr, arr[::-1].copy()]), dev), rois[:, :2], rois[:, 2])
        assert int(status.abs().sum()) == 0
        got = out[0].cpu().numpy()
        assert np.array_equal(got[:, [0, 3, 4, 5]], ref[:, [0, 3, 4, 5]]), name
        assert np.allclose(got[:, 1:3], ref[:, 1:3], rtol=1e-12, atol=0), name
        flipped = np.array([o.disk_roi_stats(arr[::-1], cx, cy, r) for cx, cy, r in rois])
        got1 = out[1].cpu().numpy()
        assert np.array_equal(got1[:, [0, 3, 4, 5]], flipped[:, [0, 3, 4, 5]])
        assert np.allclose(got1[:, 1:3], flipped[:, 1:3], rtol=1e-12, atol=0)
        per_frame = np.stack([rois[:, :2], rois[::-1, :2]])
        out2, _ = roi.disk_roi_stats_batch(T(np.stack([arr, arr]), dev), per_frame, np.stack([rois[:, 2], rois[::-1, 2]]))
        assert np.array_equal(out2[1].cpu().numpy()[::-1][:, [0, 3, 4, 5]], ref[:, [0, 3, 4, 5]])
    d = roi.DiskROI(g["slice_i16"], radius=float(rois[1, 2]), center=(float(rois[1, 0]), float(rois[1, 1])))
    assert (d.pixel_value, d.min, d.max) == tuple(g["stats_i16"][1, [5, 3, 4]])
    assert abs(d.mean - g["stats_i16"][1, 1]) < 1e-12 and abs(d.std - g["stats_i16"][1, 2]) < 1e-12
    pc = roi.DiskROI.from_phantom_center(g["slice_i16"], angle=30.0, roi_radius=7.5, dist_from_center=60.25,
                                         phantom_center=(250.3, 260.7))
    fc = g["from_center"]
    assert np.allclose(pc._xy, fc[:2], rtol=1e-15) and abs(pc.mean - fc[2]) < 1e-12 and abs(pc.std - fc[3]) < 1e-12
    assert pc.pixel_value == fc[4]
    arr = g["slice_f32"].astype(np.float64)
    boxes = np.array([[10, 40, 20, 90], [0, 512, 0, 17], [300, 301, 5, 6], [100, 228, 100, 228]], dtype=float)
    outr, st = roi.rectangle_stats_batch(T(arr[None], dev), boxes)
    assert int(st.abs().sum()) == 0
    for k, (r0, r1, c0, c1) in enumerate(boxes.astype(int)):
        v = arr[r0:r1, c0:c1]
        got = outr[0, k].cpu().numpy()
        assert got[0] == v.size and got[3] == v.min() and got[4] == v.max() and got[5] == np.median(v)
        assert abs(got[1] - v.mean()) <= 1e-12 * abs(v.mean()) and abs(got[2] - v.std()) <= 1e-12 * v.std() + 1e-15
    _, st = roi.disk_roi_stats_batch(T(arr[None], dev), [[5.0, 5.0]], 12.0)
    assert int(st[0, 0]) == 1
    with pytest.raises(IndexError):
        roi.DiskROI(arr, radius=12.0, center=(5.0, 5.0)).mean


# ----------------------------------------------------- BASELINE configs #3-#5 at batch size: properties
def _roll_batch(base: torch.Tensor, shifts):
    """[N,H,W] = base rolled by (dy, dx) per frame (wrap-around; the content sits on a uniform background)."""
    b16 = base.view(torch.int16) if base.dtype == torch.uint16 else base   # torch.roll has no uint16 kernel
    out = torch.stack([torch.roll(b16, (int(dy), int(dx)), dims=(0, 1)) for dy, dx in shifts])
    return out.view(torch.uint16) if base.dtype == torch.uint16 else out


def test_translation_equivariance_at_batch_size(dev):
    """Size-independent properties at BASELINE batch sizes (no oracle: it would take minutes): a frame translated
    by whole pixels must move every reported position by exactly that many pixels and change nothing else
    (CatPhan: to a tenth of a pixel, see below).
    config #4 WL field CAX: 512 x 1024^2 uint16;  config #5 CatPhan phantom ROI: 400 x 512^2 int16;
    config #3 picket fence: 256 x 768x1024 uint16 (picket indices and every leaf/picket position)."""
    from pylinac_amd import ct, picketfence, winston_lutz

    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    # ---- config #4: WL field centroid
    h = w = 1024
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    field = ((yy - 500).abs() < 30) & ((xx - 520).abs() < 30)
    bb = ((yy - 503) ** 2 + (xx - 517) ** 2) < 49                                   # BB: a hole in the field
    base = torch.where(bb, 12000, torch.where(field, 42000, 1500)).to(torch.int32)
    base = base + ((yy * 7 + xx * 13) % 5).to(torch.int32) * (base > 2000).to(torch.int32)
    base = base.to(torch.int16).view(torch.uint16)
    n = 512
    shifts = torch.randint(-200, 200, (n, 2), generator=g)
    shifts[0] = 0
    out = winston_lutz.field_centroids_batch(_roll_batch(base, shifts))
    ref = out[0]
    assert torch.allclose(out[:, 0], ref[0] + shifts[:, 1].to(dev, torch.float64), rtol=0, atol=1e-9)
    assert torch.allclose(out[:, 1], ref[1] + shifts[:, 0].to(dev, torch.float64), rtol=0, atol=1e-9)
    assert bool((out[:, 2] == ref[2]).all())
    # ---- config #5: CatPhan phantom ROI (scharr -> gaussian -> float Otsu -> clear_border -> fill -> label)
    hs = 512
    yy, xx = torch.meshgrid(torch.arange(hs, device=dev), torch.arange(hs, device=dev), indexing="ij")
    r = torch.hypot((yy - 250).double(), (xx - 262).double()) * 0.5
    sl = torch.full((hs, hs), -1000.0, device=dev, dtype=torch.float64)
    sl[r < 100] = 60.0
    sl[(r < 100) & (((yy // 9) + (xx // 7)) % 2 == 0)] = 95.0
    sl = sl.to(torch.int16)
    n5 = 400
    sh5 = torch.randint(-40, 40, (n5, 2), generator=g)
    sh5[0] = 0
    roi = ct.phantom_roi_batch(_roll_batch(sl, sh5), 0.5)
    # (the Otsu threshold is taken inside a disk FIXED at the image centre, ct.py:3323-3336, so a translated
    # phantom may gain or lose a few boundary pixels: the property holds to a fraction of a pixel, not exactly)
    assert (roi[:, 0] == 0).all()
    assert np.allclose(roi[:, 3], roi[0, 3] + sh5[:, 0].numpy(), rtol=0, atol=0.1)
    assert np.allclose(roi[:, 4], roi[0, 4] + sh5[:, 1].numpy(), rtol=0, atol=0.1)
    assert np.allclose(roi[:, 2], roi[0, 2], rtol=5e-3)
    # ---- config #3: picket fence (AS1000 geometry), pickets translated along x
    hp, wp, dpmm = 768, 1024, 1 / 0.390625
    xs = torch.arange(wp, device=dev, dtype=torch.float64)
    prof = torch.zeros(wp, device=dev, dtype=torch.float64)
    for k in range(10):
        prof += torch.exp(-0.5 * ((xs - (180 + k * 15 * dpmm + (k % 3) * 0.37)) / 3.1) ** 2)
    frame = (2000 + 50000 * prof)[None, :].expand(hp, wp)
    frame = (frame + ((torch.arange(hp, device=dev)[:, None] * 3 + torch.arange(wp, device=dev)[None, :]) % 7)).round()
    frame = frame.to(torch.int32).to(torch.int16).view(torch.uint16).contiguous()
    n3 = 256
    dx = torch.randint(-60, 60, (n3,), generator=g)
    dx[0] = 0
    # roll along x only; the additive pattern above is NOT rolled with it -> rebuild per frame from the rolled profile
    frames = torch.stack([torch.roll(frame.view(torch.int16), int(d), dims=1) for d in dx]).view(torch.uint16)
    res = picketfence.analyze_batch(frames, dpmm, num_pickets=10)
    assert bool((res.picket_count == 10).all())
    assert torch.equal(res.picket_idx[:, :10] - res.picket_idx[0:1, :10], dx.to(dev, torch.int32)[:, None].expand(-1, 10))
    pos0 = res.position[0:1]
    ok = torch.isfinite(pos0).expand_as(res.position)
    assert bool(ok[0].any()) and torch.equal(torch.isfinite(res.position), ok)
    diff = (res.position - pos0 - dx.to(dev, torch.float64)[:, None, None])[ok]
    assert float(diff.abs().max()) < 1e-9


def test_single_profile_inflection_derivative_vs_reference(golden, dev):
    """Edge.INFLECTION_DERIVATIVE on the device (Gaussian smoothing, np.gradient, peak / valley search of the
    gradient): inflection indices exact, values and every field_data scalar to 1e-9, protocol metrics to 1e-9
    of the reference run on its 20 frozen profiles."""
    from pylinac_amd import profile as pp
    from tests.test_oracle_golden import _sp_calculators, _sp_check

    g = golden("single_profile")
    calcs = _sp_calculators()
    for i in range(20):
        for mode, interp in (("none", None), ("linear", "Linear")):
            p = pp.SingleProfile(g[f"fx{i}.y"], x_values=g[f"fx{i}.x"], interpolation=interp,
                                 edge_detection_method="Inflection Derivative")
            _sp_check(g, f"fx{i}.infl_{mode}", p, calcs, vtol=0, ftol=1e-9)
            inf = p.inflection_data()
            ref = g[f"fx{i}.infl_{mode}.infl"]
            got = np.array([inf[str(k)] for k in g["infl_keys"]])
            assert np.array_equal(got[:2], ref[:2]) and np.allclose(got[2:], ref[2:], rtol=1e-9, atol=1e-12), (i, mode)
    with pytest.raises(ValueError):
        pp.SingleProfile(g["fx0.y"], x_values=g["fx0.x"], interpolation=None).inflection_data()
    # (the Hill-fit edge method: test_single_profile_hill_and_penumbra_vs_reference_golden)


# ------------------------------------------------------------------------------------ 2-D gamma (f4)
def test_gamma_2d_vs_reference_golden(golden, dev):
    """pl_gamma2d against the reference's own gamma_2d: bit-identical gamma maps (NaN positions included) for its
    known-answer inputs and for dose-like images (DTA 1-4, global / local dose, thresholds, fill values, NaNs in
    the evaluation); batch form = per-pair results; same ValueError for non-2-D input."""
    from pylinac_amd import gamma as pg
    from tests.test_oracle_golden import _gamma_cases

    g = golden("gamma")
    for k, ref, ev, kw, want in _gamma_cases(g):
        got = pg.gamma_2d(ref, ev, device=dev, **kw).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True), (k, kw)
    _, ref, ev, kw, want = list(_gamma_cases(g))[11]
    both = pg.gamma_2d(np.stack([ref, ref[::-1].copy()]), np.stack([ev, ev[::-1].copy()]), device=dev, **kw).cpu().numpy()
    assert np.array_equal(both[0], want, equal_nan=True)
    assert np.array_equal(both[1], o.gamma_2d(ref[::-1], ev[::-1], **kw), equal_nan=True)
    with pytest.raises(ValueError):
        pg.gamma_2d(np.ones(5), np.ones((5, 5)), device=dev)
    # full-size property (no oracle): gamma of an image against itself is 0 wherever it is evaluated
    big = torch.rand((4, 1024, 1024), device=dev, dtype=torch.float64) + 0.5
    z = pg.gamma_2d(big, big, distance_to_agreement=3)
    assert float(z.max()) == 0.0


def test_gamma_1d_vs_reference_golden(golden, dev):
    """pl_gamma1d against the reference's own gamma_1d (known-answer inputs, non-uniform / reversed evaluation
    abscissae, local dose, fractional DTA): sample positions and sampled evaluation values bit-identical; gamma
    within 2 ulp -- the reference squares with Python's ``float ** 2`` (libm pow), which is not always the
    correctly rounded product x*x the device (and numpy) computes: 1 value in ~2000 differs in the last bit.
    Also the reference's ValueErrors."""
    from pylinac_amd import gamma as pg
    from tests.test_oracle_golden import _gamma1d_cases

    g = golden("gamma1d")
    for k, kw, want in _gamma1d_cases(g):
        got = pg.gamma_1d(device=dev, **kw)
        assert np.array_equal(np.isnan(got[0]), np.isnan(want[0]))
        assert np.allclose(got[0], want[0], rtol=4.5e-16, atol=0, equal_nan=True), k
        assert np.array_equal(got[1], want[1], equal_nan=True) and np.array_equal(got[2], want[2]), k
    with pytest.raises(ValueError):
        pg.gamma_1d(np.ones((2, 2)), np.ones(4), device=dev)
    with pytest.raises(ValueError):
        pg.gamma_1d(np.ones(5), np.ones(5), resolution_factor=1.5, device=dev)
    with pytest.raises(ValueError):
        pg.gamma_1d(np.ones(5), np.ones(5), reference_coordinates=np.arange(5.0) + 10, device=dev)


def test_picket_fence_other_leaf_banks_vs_reference_golden(golden, dev):
    """HD_MILLENNIUM / AGILITY / HALCYON_DISTAL / BMOD banks (and AGILITY LEFT_RIGHT) against the reference's own analyze()"""
    import next_row_checks as checks

    checks.check_pf_mlc_device(golden("picketfence_mlc"), dev)


def test_gamma_geometric_vs_reference_golden(golden, dev):
    import next_row_checks as checks

    checks.check_gamma_geometric(golden, dev)


def test_median3_eight_columns_per_lane_shapes(dev):
    """median3_oct_kernel (width % 8 == 0, 16-byte aligned frames): one lane, partial waves, several waves per row
    (neighbour columns fetched across a wave boundary), heights that are not a multiple of the 16-row group, uint16
    and int16 -- against scipy's median_filter via the oracle; other widths take the pair / scalar kernels."""
    from pylinac_amd import ops

    rng = np.random.default_rng(17)
    for shape in [(2, 5, 8), (1, 33, 16), (2, 40, 520), (1, 70, 1032), (1, 17, 2048), (3, 2, 512), (1, 16, 504)]:
        a = rng.integers(0, 65536, shape, dtype=np.uint16)
        a[0, : shape[1] // 2] = (a[0, : shape[1] // 2] // 4096) * 4096            # plateaus: many ties
        for arr in (a, (a.astype(np.int32) - 32768).astype(np.int16)):
            ref = np.stack([o.filter(f, 3, "median") for f in arr])
            assert np.array_equal(ops.median_filter(T(arr, dev), 3).cpu().numpy(), ref), (shape, arr.dtype)


# ------------------------------------------------------------------------------------ XIM decode (f1)
def test_xim_reader_vs_reference_reader(golden, dev, tmp_path):
    """pylinac_amd.xim.XIM (pl_xim_decode) on synthetic compressed .xim files against what the reference's own XIM
    reader decoded from the same bytes: pixel arrays bit-identical (int32 / int16, 1/2/4-byte differences,
    wrap-around, a 2-row image), header fields, histogram, typed properties, dpmm; then the size-independent
    property at the real detector size: encode -> decode round trip of a 1280 x 1280 image."""
    import json

    from pylinac_amd import xim as px
    from tests.test_oracle_golden import _xim_split

    g = golden("xim")
    for name in "abcd":
        path = tmp_path / f"{name}.xim"
        path.write_bytes(g[f"{name}.file"].tobytes())
        x = px.XIM(str(path), device=dev)
        want = g[f"{name}.array"]
        assert x.array.cpu().numpy().dtype == want.dtype and np.array_equal(x.array.cpu().numpy(), want), name
        assert (x.img_height_px, x.img_width_px) == want.shape and x.format_id == "VMS.XI"
        assert x.dpmm == float(g[f"{name}.dpmm"]) and np.array_equal(x.histogram, g[f"{name}.histogram"])
        props = json.loads(str(g[f"{name}.props"]))
        assert set(props) == set(x.properties)
        for k, v in props.items():
            assert np.array_equal(np.asarray(x.properties[k]), np.asarray(v)), k
        w, h, bpp, lut, buf = _xim_split(g[f"{name}.file"])
        assert np.array_equal(px.decode_xim_pixels(lut, buf, w, h, bpp, device=dev).cpu().numpy(), want)
    bad = g["a.file"].copy()
    w, h, bpp, lut, buf = _xim_split(bad)
    lut = lut.copy()
    lut[5] = 0xFF                                   # size code 3: the reference raises KeyError(3)
    with pytest.raises(KeyError):
        px.decode_xim_pixels(lut, buf, w, h, bpp, device=dev)
    with pytest.raises(ValueError):
        px.decode_xim_pixels(lut, buf, w, h, 3, device=dev)
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:1280, 0:1280]
    img = (30000 + 20000 * np.sin(yy / 97.0) * np.cos(xx / 131.0) + rng.normal(0, 50, (1280, 1280))).round().astype(np.int64)
    img.ravel()[rng.integers(0, img.size, 50)] = 1 << 21
    lut, buf = o.xim_encode(img)
    got = px.decode_xim_pixels(lut, buf, 1280, 1280, 4, device=dev).cpu().numpy()
    assert np.array_equal(got, img.astype(np.int32))


# ----------------------------------------------------------------------------------------- Canny (f2)
def test_canny_vs_skimage_golden(golden, dev):
    """Device canny against scikit-image 0.18.3's own feature.canny with pylinac's parameters (sigma 2 / 4, quantile
    thresholds), absolute thresholds, large-valued float images, an empty result: identical edge maps; batch form;
    the label count of the edge map (what planar_imaging.py:585-587 consumes) equals scipy's."""
    from scipy import ndimage as ndi

    from pylinac_amd import canny as pc
    from pylinac_amd import ops

    g = golden("canny")
    from tests.test_oracle_golden import _canny_cases

    for k, img, kw, want in _canny_cases(g):
        got = pc.canny(img, device=dev, **kw)
        assert got.shape == want.shape
        diff = int((got.cpu().numpy().astype(bool) != want).sum())
        assert diff == 0, (k, kw, diff)
    _, img, kw, want = next(iter(_canny_cases(g)))
    both = pc.canny(np.stack([img, img[::-1].copy()]), device=dev, **kw).cpu().numpy().astype(bool)
    assert np.array_equal(both[0], want) and np.array_equal(both[1], o.canny(img[::-1], **kw))
    labels, num = ops.label(T(want.astype(np.uint8)[None], dev), 8)
    assert int(num[0]) == ndi.label(want, np.ones((3, 3)))[1]
    with pytest.raises(ValueError):
        pc.canny(img, low_threshold=1.5, use_quantiles=True, device=dev)
    with pytest.raises(TypeError):
        pc.canny(img.astype(np.float32), device=dev)



def test_hough_line_vs_skimage_golden(golden, dev):
    """pl_hough_line against scikit-image 0.18.3's transform.hough_line: identical accumulators, angles and bins."""
    from pylinac_amd import canny as pc

    g = golden("hough")
    for k in range(3):
        theta = None if k == 0 else g[f"a{k}"]
        acc, a, d = pc.hough_line(g[f"img{k}"], theta, device=dev)
        assert np.array_equal(acc.cpu().numpy().astype(np.uint64), g[f"h{k}"]), k
        assert np.array_equal(a, g[f"a{k}"]) and np.array_equal(d, g[f"d{k}"])
    with pytest.raises(ValueError):
        pc.hough_line(np.zeros(5), device=dev)


def test_hough_line_peaks_vs_skimage_golden(golden, dev):
    """f2 (second half): planar.hough_line_peaks (pl_max_filter1d x2, pl_peak_candidates, pl_label + the host's greedy
    walk) against scikit-image 0.18.3's transform.hough_line_peaks: identical heights, angles and distances."""
    import next_row_checks as checks

    checks.check_hough_line_peaks(golden, dev)


def test_region_moments_vs_skimage_regionprops_golden(golden, dev):
    """f2 / a16: exact integer raw moments per label (pl_region_moments) and scikit-image 0.18.3's centroid /
    inertia_tensor / orientation / eccentricity formed from them, on 19 regions including axis-swap-symmetric ones."""
    import next_row_checks as checks

    checks.check_region_moments_kernel(golden, dev)


def test_phantom_outline_vs_skimage_golden(golden, dev):
    """f2 (second half): canny -> label -> bbox table -> phantom_ski_region -> region.image -> hough_line on three
    synthetic phantom frames against scikit-image's own canny / label / regionprops / hough_line."""
    import next_row_checks as checks

    checks.check_phantom_outline(golden, dev)
    from pylinac_amd import planar

    g = golden("planar")
    with pytest.raises(ValueError, match="Unable to find the phantom"):
        planar.find_phantom_region(torch.from_numpy(g["sq0.img"]).to(dev), 10.0, sigma=4)


def test_max_filter1d_vs_scipy(dev):
    """pl_max_filter1d == ndimage.maximum_filter1d(mode='constant', cval=0) on int64 / float64 / uint16 frames with
    negative values (the zero padding wins at the borders), windows wider than the frame, both axes."""
    from scipy import ndimage as ndi

    from pylinac_amd import planar

    rng = np.random.default_rng(77)
    for arr in (rng.integers(-50, 50, (2, 37, 29)).astype(np.int64), rng.normal(size=(1, 40, 33)),
                rng.integers(0, 65535, (1, 21, 64)).astype(np.uint16)):
        for axis in (0, 1):
            for half in (0, 1, 5, 70):
                got = planar.max_filter1d(T(arr, dev), half, axis).cpu().numpy()
                want = np.stack([ndi.maximum_filter1d(f, size=2 * half + 1, axis=axis, mode="constant", cval=0) for f in arr])
                assert got.dtype == want.dtype and np.array_equal(got, want), (arr.dtype, axis, half)


def test_single_profile_hill_and_penumbra_vs_reference_golden(golden, dev):
    """f4 (second half): SingleProfile with Edge.INFLECTION_HILL (device smoothing / gradient / peak searches /
    resampling; the four-parameter Hill fits by scipy's curve_fit on the host, like the reference) and penumbra() for the
    three edge methods against the reference's own numbers (tests/golden/hill.npz: 132 profiles + the ones it rejects).
    1e-9 relative; 1e-5 for everything downstream of a Hill fit (MINPACK stops at 1.5e-8 and numpy's vectorised pow is
    not bit-reproducible run to run: the reference reproduces its own fitted parameters only to ~1e-7)."""
    import warnings

    import next_row_checks as checks
    from pylinac_amd import profile

    def make(values, edge, **kw):
        return profile.SingleProfile(values, edge_detection_method=edge, **kw)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert checks.check_hill_and_penumbra(golden("hill"), make, tol=1e-9, spline_tol=1e-6) == 132


def test_starshot_vs_reference_golden(golden, dev):
    """Starshot.analyze (device: percentiles, inversion, grounding, axis maxima, 20-radius circle gather, 1-D Gaussian,
    FWXM peaks; host: line pairing, Nelder-Mead wobble, retry sweep) against the reference's own Starshot on six
    synthetic frames: start point, star profile, peaks, lines to 1e-9; wobble to 1e-7; angles; pass flag."""
    import next_row_checks as checks
    from pylinac_amd import starshot

    assert checks.check_starshot(golden("starshot"), lambda f, dpi, sid: starshot.Starshot(f, dpi=dpi, sid=sid)) == 6
    with pytest.raises(ValueError, match="DPI"):
        starshot.Starshot(np.zeros((10, 10), np.uint16))
    s = starshot.Starshot(golden("starshot")["four.frame"].copy(), dpi=100, sid=1000)
    with pytest.raises(ValueError):
        s.analyze(radius=0.1)


@pytest.mark.gpu
def test_hist16_wl_more_frames_than_cus(dev):
    """pl_hist16_wl's two-workgroups-per-CU instantiation (9 728-bin windows; taken for batches of more frames than CUs) against
    pl_hist16 / pl_order_stats / pl_edge_minmax and np.bincount: 260 frames 512 x 512, uint16 and int16."""
    import next_row_checks as checks

    assert checks.check_hist16_wl_many_frames(dev) == 520


@pytest.mark.gpu
def test_circle_profile_ring_vs_gathers(dev):
    """pl_circle_profile_ring == pl_circle_profile_combined_ex bit for bit (samples, margins): borders, outside / NaN centres, four
    dtypes, k = 0 .. 3, honest / too narrow / too wide promises about the radii."""
    import next_row_checks as checks

    assert checks.check_circle_profile_ring(dev) == 84


@pytest.mark.gpu
def test_starshot_analyze_batch(golden, dev):
    """starshot.analyze_batch: four stacks (golden frame + shifted / inverted copies) against the reference's own numbers for
    the leading frame and the class API for every frame; per-frame status codes for the cases the class raises on."""
    import next_row_checks as checks

    assert checks.check_starshot_batch(golden("starshot"), dev) == 12





def test_rectangle_roi_vs_reference_golden(golden, dev):
    """f3 (second half): pl_polygon_roi_stats / RectangleROI against the reference's own RectangleROI (rotated,
    unrotated, clipped) and scikit-image 0.18.3's draw.polygon pixel sets."""
    import next_row_checks as checks

    checks.check_rectangle_roi(golden, dev)


def test_image_gamma_bakai_vs_reference_golden(golden, dev):
    """ArrayImage.gamma (pl_bakai_mask, float32 pl_sobel, pl_bakai_gamma, exact float64 percentiles for the inversion
    check) against the reference's own ArrayImage.gamma: identical float64 maps incl. the NaN pattern, for uint16 and
    float64 inputs and every option; the reference's AttributeErrors for mismatched DPI / size."""
    from pylinac_amd.image import ArrayImage
    from tests.test_oracle_golden import _bakai_cases

    g = golden("bakai")
    for name, ref, cmp_, kw, want in _bakai_cases(g):
        got = ArrayImage(ref.copy(), dpi=75.6).gamma(ArrayImage(cmp_.copy(), dpi=75.6), **kw)
        assert got.dtype == np.float64 and np.array_equal(np.isnan(got), np.isnan(want)), name
        assert np.array_equal(got, want, equal_nan=True), (name, float(np.nanmax(np.abs(got - want))))
    a = ArrayImage(g["u16.ref"].copy(), dpi=75.6)
    with pytest.raises(AttributeError):
        a.gamma(ArrayImage(g["u16.cmp"].copy(), dpi=100))
    with pytest.raises(AttributeError):
        a.gamma(ArrayImage(g["u16.cmp"][:-5].copy(), dpi=75.6))


# --------------------------------------------------------------------- percentile-driven decisions (a6)
def test_image_decisions_vs_oracle(dev):
    """has_noise / pf_orientation / corners_inverted (batched) and clean_edges against the oracle restatements of
    PFDicomImage._has_noise, PicketFence.orientation, BaseImage.check_inversion and WLBaseImage._clean_edges
    (themselves checked against the reference's methods in tests/test_oracle_vs_reference.py): frames with and
    without dead/hot pixels, picket-like stripes along either axis, an inverted frame, noisy edges to crop."""
    from pylinac_amd import decisions as dc
    from tests.test_oracle_vs_reference import _decision_frames

    frames = _decision_frames()
    for shape in {f.shape for f in frames}:
        group = [f for f in frames if f.shape == shape]
        t = T(np.stack(group), dev)
        assert list(dc.has_noise(t)) == [o.has_noise(a) for a in group]
        assert dc.pf_orientation(t) == [o.pf_orientation(a) for a in group]
        assert list(dc.corners_inverted(t)) == [o.corners_inverted(a) for a in group]
        assert list(dc.corners_inverted(t, box_size=10, position=(0.1, 0.2))) == \
            [o.corners_inverted(a, 10, (0.1, 0.2)) for a in group]
    for a in frames[:3]:
        b = a.copy()
        b[0, :40] = 65535
        b[:, -1] = 0
        b[-2:, 10:30] = 65535
        want = o.clean_edges(b)
        got = dc.clean_edges(T(b, dev)).cpu().numpy()
        assert want.shape != b.shape and np.array_equal(got, want)
        assert np.array_equal(dc.clean_edges(T(a, dev)).cpu().numpy(), o.clean_edges(a))


# ----------------------------------------------------------------------------------------------------
# Added after the last GPU session of round 1: checked against the emulated device (tests/emu_backend.py) and the golden
# vectors, first hardware run = the round-end suite.  Kept at the end of the file so that `-x` cannot hide the tests above.
# ----------------------------------------------------------------------------------------------------
def test_contrast_rois_vs_reference_golden(golden, dev):
    """f3: LowContrastDiskROI / HighContrastDiskROI (device ROI statistics + pylinac.core.contrast formulas) against the
    reference's own classes for the four two-element contrast algorithms."""
    import next_row_checks as checks

    checks.check_contrast_rois(golden, dev)


def test_canny_integer_images_vs_skimage_golden(golden, dev):
    """canny on uint8 / uint16 / int16 images: identical edge maps to scikit-image 0.18.3 (img_as_float scaling)."""
    import next_row_checks as checks

    checks.check_canny_integer_images(golden, dev)


def test_rescale_dicom_values_vs_oracle(dev):
    """f1 (DICOM half; parity UNPINNED -- pydicom is absent from the build container): device rescale / inversion against
    the oracle's restatement of pydicom's apply_rescale + the reference's own inversion expression."""
    import next_row_checks as checks

    checks.check_rescale_dicom_values(dev)


def test_thickness_roi_vs_reference_golden(golden, dev):
    """f3: ThicknessROI (CatPhan slice-thickness ramps) against the reference's own pylinac.ct.ThicknessROI."""
    import next_row_checks as checks

    checks.check_thickness_roi(golden, dev)


def test_field_strips_vs_reference_golden(golden, dev):
    """a7: FieldAnalysis strip profiles (`np.mean(array[bottom:top, :], 0)` and the vertical twin, with the reference's
    edge rounding / clipping) and its centre search (axis sums -> SingleProfile) against the reference's own methods."""
    import next_row_checks as checks

    checks.check_field_strips(golden, dev)


def test_edge_profiles_vs_reference_golden(golden, dev):
    """f4: InflectionDerivativeProfile / HillProfile (profile.py:612-740; device smoothing, gradient and spline solve,
    host BFGS / curve_fit like the reference) against the reference's own classes on its 20 frozen profiles, an EPID
    profile and FFF-style profiles: edges / centre / width to 1e-5 of the profile extent, geometric centre and CAX index
    to 1e-9, and the two profiles the reference rejects."""
    import warnings

    import next_row_checks as checks
    from pylinac_amd import profile

    def make(kind, values, **kw):
        cls = profile.HillProfile if kind == "hill" else profile.InflectionDerivativeProfile
        return cls(values, **kw)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert checks.check_edge_profiles(golden("edge_profiles"), make) == 65
        checks.check_edge_profile_known_answers(make)


def test_catphan_volume_localisation_vs_reference_golden(golden, dev):
    """Config #5's loop over slices: find_phantom_axis (phantom ROI of every slice -> outlier screen -> linear fits) and
    find_origin_slice (collapsed circle profile through the HU inserts + percentile test on every second slice) against
    the reference's own CatPhanBase methods on two synthetic tilted volumes; find_phantom_roll (air-bubble regions by
    filled area and eccentricity) to 1e-9 degrees."""
    import next_row_checks as checks

    checks.check_catphan_volume(golden, dev)


def test_profile_base_fields_vs_reference_golden(golden, dev):
    """ProfileBase.field_x_values / field_values / field_indices / resample_to (profile.py:299-352, 392-431) through
    FWXMProfile: the reference's known answers and its own results on five frozen profiles."""
    import next_row_checks as checks

    checks.check_profile_base_fields(golden("edge_profiles"))


def test_as_resampled_vs_scipy_zoom(dev):
    """ProfileBase.as_resampled (pl_zoom1d_cubic) against scipy.ndimage.zoom(order=3, mode="nearest", grid_mode=False):
    1e-12 on random profiles of six lengths x six factors, the reference's length / range / type known answers, and the
    integer-dtype rounding."""
    import next_row_checks as checks

    checks.check_as_resampled(dev)


def test_bit_invert_and_convert_to_dtype(dev):
    """a5 companions (array_utils.py:80-89, 171-198): the reference's literal known answers and numpy's expressions."""
    import next_row_checks as checks

    checks.check_bit_invert_and_convert_to_dtype()


def test_rotate_matches_skimage(dev, golden):
    """BaseImage.rotate (image.py:780-783): bit-exact on skimage 0.18.3's own inverse map, 1e-12 through the public method."""
    import next_row_checks as checks

    checks.check_rotate(golden, dev)


def test_large_rois_stream(dev):
    """f3: ROIs beyond the 16384-pixel LDS buffer (streaming passes) and the pixel-based out-of-frame decision."""
    import next_row_checks as checks

    checks.check_large_rois(dev)


def test_ground_promotes_like_numpy(dev):
    """array_utils.ground (pylinac/core/array_utils.py:92-102 = ``array - array.min() + value``): numpy's result dtype for
    every kind of ``value`` (a Python float 0.0 / 1.0 promotes an integer array too; inf / nan do not raise)."""
    from pylinac_amd import array_utils as au

    import next_row_checks as checks

    checks.check_ground_promotion()
    f = np.array([1.5, -2.0, 4.0])
    assert np.array_equal(au.ground(f, 1), f - f.min() + 1)


def test_bench_line_contract(dev):
    """bench.py prints ONE JSON line with the driver's keys: metric / value / unit / n_gpus / steps / warmup / ms_per_step /
    higher_is_better / scaling / vs_baseline / dtype / data / config.workload, the roofline object of the dominant kernel
    (bound, achieved, peak, unit, frac, traffic) and the cpu_baseline object (value, unit, cores, kind, sample).  A short run
    on small frames (the default size is what the driver times)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "8", "--height", "256", "--width", "256",
                        "--steps", "3", "--warmup", "1", "--no-configs", "--cpu-frames", "2"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "images/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0
    assert d["parity_sample"]["ok"] is True and d["parity_sample"]["units"] >= 1



@pytest.mark.gpu
def test_mask_regions_vs_scipy(dev):
    """pl_mask_regions (one workgroup per frame: bit plane + row runs in LDS) against scipy's clear_border / fill_holes /
    label / regionprops sums, incl. a crowded 512 x 512 frame that must report status 1."""
    import next_row_checks as checks

    checked, overflowed = checks.check_mask_regions(dev, shapes=((64, 64), (70, 130), (33, 65), (17, 5), (96, 192), (1, 1), (40, 64),
                                                                 (300, 300), (512, 512)))
    # 9 shapes x 7 masks x up to 5 settings; the speckle masks of the two large shapes overflow the run list (status 1)
    assert checked >= 200 and overflowed >= 10 and checked + overflowed >= 260, (checked, overflowed)


@pytest.mark.gpu
def test_scharr_gaussian_bit_identical(dev):
    import next_row_checks as checks

    checks.check_scharr_gaussian(dev, shapes=((2, 70, 130, np.int16), (1, 33, 65, np.uint16), (2, 32, 64, np.int16),
                                              (1, 5, 7, np.int16), (1, 100, 9, np.uint16), (3, 512, 512, np.int16)))


@pytest.mark.gpu
def test_edge_otsu_one_launch(dev):
    import next_row_checks as checks

    checks.check_edge_otsu(dev, shapes=((2, 70, 130, np.int16), (1, 33, 65, np.uint16), (1, 100, 9, np.uint16),
                                        (2, 64, 200, np.int16), (3, 512, 512, np.int16)))


@pytest.mark.gpu
def test_circle_profile_combined(dev):
    import next_row_checks as checks

    checks.check_circle_profile_combined(dev)
    checks.check_circle_profile_combined(dev, n_volumes=3, spv=20, h=256, w=256)


@pytest.mark.gpu
def test_phantom_roi_fused_vs_separate(dev):
    import next_row_checks as checks

    checks.check_phantom_roi_fused_vs_separate(dev, slices=tuple(range(0, 80, 4)))


@pytest.mark.gpu
def test_histogram16_one_read_vs_bincount(dev):
    """pl_hist16's single-read kernel (two LDS windows, hot-value peel, global atomics between the windows) == np.bincount:
    bimodal / clipped-noise / narrow / uniform-random / constant frames, uint16 and int16, sizes off the 8-pixel vectors."""
    import next_row_checks as checks

    assert checks.check_histogram16_one_read(dev) == 64


@pytest.mark.gpu
def test_fused_tail_vs_separate(dev):
    """pl_median3_threshold_colparts_u16 + pl_colparts_profile_fwxm == the four separate launches they replace, bit for bit
    (partial bands, partial column groups, a frame wholly below / above its threshold), plus one 1024 x 1024 batch."""
    import next_row_checks as checks

    assert checks.check_fused_tail_vs_separate(dev, shapes=((3, 200, 520), (2, 128, 64), (2, 130, 1032), (1, 2, 8), (3, 1024, 1024))) == 5


@pytest.mark.gpu
def test_canny_masked_vs_skimage_golden(golden, dev):
    """canny(mask=...) == scikit-image 0.18.3's feature.canny with the same mask (golden from the helper interpreter)."""
    import next_row_checks as checks

    checks.check_canny_masked(golden, dev)


@pytest.mark.gpu
def test_hill_fit_matches_scipy(dev):
    """pl_hill_fit against scipy's leastsq (what the reference's Hill.fit calls) on 400 synthetic penumbra windows."""
    import next_row_checks as checks
    from pylinac_amd import ops

    def fit(xs, ys, lens):
        p, info, nfev = ops.hill_fit(T(xs, dev), T(ys, dev), T(lens, dev))
        return p.cpu().numpy(), info.cpu().numpy(), nfev.cpu().numpy()

    def fit_ex(xs, ys, lens):                               # + the last accepted step: settled fits are held to 1e-5, all of them
        p, info, nfev, step = ops.hill_fit(T(xs, dev), T(ys, dev), T(lens, dev), last_step=True)
        return p.cpu().numpy(), info.cpu().numpy(), nfev.cpu().numpy(), step.cpu().numpy()

    assert checks.check_hill_fit_vs_scipy(fit_ex, n=400, seed=3) >= 360
    assert checks.check_hill_fit_kernels_agree(fit, n=512) >= 460
    assert checks.check_hill_fit_pathological(fit, fit_ex)


@pytest.mark.gpu
def test_hill_batch_vs_reference_golden_and_single(golden, dev):
    """single_profile_hill_batch: every INFLECTION_HILL profile of hill.npz against the reference's own numbers, then 2048
    synthetic profiles in one batch against the per-profile mirror (scipy's curve_fit on the host) on a sample."""
    import time
    import warnings

    import next_row_checks as checks
    from pylinac_amd import profile

    batch = profile.single_profile_hill_batch

    assert checks.check_hill_batch(golden("hill"), batch) == 46
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = checks.check_hill_batch_vs_single(
            batch, lambda v, **kw: profile.SingleProfile(v, edge_detection_method=profile.Edge.INFLECTION_HILL, **kw),
            n=2048, length=200, sample=range(0, 2048, 256))
    info = res.info.cpu().numpy()
    assert ((info >= 1) & (info <= 4)).all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert checks.check_hill_batch_options(
            batch, lambda v, **kw: profile.SingleProfile(v, edge_detection_method=profile.Edge.INFLECTION_HILL, **kw)) == 36
    profs = T(checks.beam_profiles(2048, 200), dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    profile.single_profile_hill_batch(profs)
    torch.cuda.synchronize()
    print(f"\nsingle_profile_hill_batch: 2048 profiles x 200 detectors (resampled x10), BEAM_CENTER normalisation: "
          f"{(time.perf_counter() - t0) * 1e3:.2f} ms")


@pytest.mark.gpu
def test_fwhm_batch_vs_single(golden, dev):
    """single_profile_fwhm_batch against the reference's own SingleProfile numbers (20 frozen profiles x 3 resampling modes + the
    EPID options of single_profile.npz), against the per-profile SingleProfile for every normalisation / interpolation choice,
    then 4096 profiles in one batch: a sample against the mirror, all rows with a peak."""
    import next_row_checks as checks
    from pylinac_amd import profile

    assert checks.check_fwhm_batch_golden(golden("single_profile"), profile.single_profile_fwhm_batch) == 66
    fns = dict(infl=profile.single_profile_inflection_batch, fwhm=profile.single_profile_fwhm_batch,
               hill=profile.single_profile_hill_batch)
    assert checks.check_profile_batch_golden(golden("profile_batch"), fns) == 33      # the reference's own numbers, three methods

    assert checks.check_fwhm_batch(profile.single_profile_fwhm_batch, lambda v, **kw: profile.SingleProfile(v, **kw)) == 108
    assert checks.check_inflection_batch(
        profile.single_profile_inflection_batch,
        lambda v, **kw: profile.SingleProfile(v, edge_detection_method=profile.Edge.INFLECTION_DERIVATIVE, **kw)) == 36
    profs = checks.beam_profiles(4096, 200, seed=3)
    res = profile.single_profile_fwhm_batch(T(profs, dev))
    d = {k: v.cpu().numpy() for k, v in res.fwxm_data(50).items()}
    assert (d["peaks"] == 1).all()
    for i in range(0, 4096, 512):
        want = profile.SingleProfile(profs[i].copy()).fwxm_data(50)
        for k in ("left index (exact)", "right index (exact)", "center value (@rounded)", "width (exact)"):
            assert np.isclose(d[k][i], want[k], rtol=1e-9, atol=1e-9), (i, k)


@pytest.mark.gpu
def test_dicom_decode_vs_fixtures_and_frombuffer(golden, dev):
    """f1, the DICOM half (pylinac/core/image.py:1431-1444): Part-10 fixtures (tests/golden/make_dicom_golden.py), every
    container format / byte order / alignment against np.frombuffer, and a batch at the bench's frame size: 64 files of one
    1024 x 1024 uint16 frame each, file bytes uploaded as they are, one launch -> the frames that were encoded."""
    import importlib.util

    import next_row_checks as checks
    from pylinac_amd import dicom

    checks.check_dicom_golden(golden, dev)
    checks.check_dicom_decode_fuzz(dev, frame_shapes=((9, 14), (16, 16), (61, 67)), n=3)
    spec = importlib.util.spec_from_file_location("make_dicom_golden", os.path.join(os.path.dirname(__file__), "golden", "make_dicom_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = np.random.default_rng(5)
    arrays = [rng.integers(0, 65536, (1024, 1024), dtype=np.uint16) for _ in range(64)]
    files = [gen.part10(a, pad_text="x" * (2 * (k % 3))) for k, a in enumerate(arrays)]     # Pixel Data at varying alignments
    frames, metas = dicom.load_frames(files, device=dev)
    assert frames.shape == (64, 1024, 1024) and frames.dtype == torch.uint16
    assert np.array_equal(dicom._to_numpy(frames), np.stack(arrays))
    assert len({m.PixelData[0] % 4 for m in metas}) == 2


@pytest.mark.gpu
def test_epid_run_from_host_equals_run(dev):
    """EpidPipeline.run_from_host (pinned host frames, copy of piece k + 1 under the kernels of piece k) returns exactly what
    run() returns on the same frames, for every chunking, pass after pass (the staging buffer is reused)."""
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import epid_open_field_frames

    n, h, w = 12, 256, 320
    frames = epid_open_field_frames(n, h, w, seed0=21, device=dev, field_mm=40.0)
    host = frames.cpu().pin_memory()
    pipe = EpidPipeline(n, h, w, dev)
    ref = pipe.run(frames)
    want = (ref.frames.clone(), ref.profile.clone(), ref.record().clone())
    for chunks in (1, 2, 5, 12, 40):
        for _ in range(2):
            res = pipe.run_from_host(host, chunks)
            torch.cuda.synchronize()
            assert torch.equal(res.frames, want[0]) and torch.equal(res.profile, want[1])
            assert torch.equal(torch.nan_to_num(res.record()), torch.nan_to_num(want[2]))


@pytest.mark.gpu
def test_ctp528_device_axis_path(dev):
    """Round 6: the circle profiles of ct.ctp528_batch are placed by a fit made on the device and verified against the
    reference's np.polyfit through each profile's decision margin (tests/next_row_checks.py)."""
    import next_row_checks as checks

    checks.check_ctp528_device_axis_path(dev, n_slices=12, size=512, mmpp=0.5)


@pytest.mark.gpu
def test_analyze_batch_takes_what_the_loader_produces(golden, dev):
    """the batched Winston-Lutz and picket-fence analyzers on int16 and float64 frames (post
    apply_rescale), against the oracle on exactly those arrays and against the uint16 result; 1024 x 1024 frames included."""
    import next_row_checks as checks
    from pylinac_amd import picketfence as ppf
    from pylinac_amd import winston_lutz as wl
    from pylinac_amd.synthetic import pf_frames, wl_frames

    checks.check_wl_analyze_batch_other_dtypes(golden, dev, frames=(0, 6, 7))
    checks.check_pf_other_dtypes(golden("picketfence_mlc"), dev)
    # BASELINE's frame sizes: 1024 x 1024 Winston-Lutz frames as rescaled float64, 768 x 1024 picket fences as float64 / int16
    fr = torch.from_numpy(wl_frames(6)).to(dev)
    base = wl.analyze_batch(fr, 1 / 0.336, 5.0)
    scaled = fr.to(torch.float64) * 4.315e-5 - 0.25
    res = wl.analyze_batch(scaled, 1 / 0.336, 5.0)
    assert np.array_equal(res["record"][:, :2], base["record"][:, :2]) and np.array_equal(res["status"], base["status"])
    assert np.allclose(res["record"][:, 2:], base["record"][:, 2:], rtol=0, atol=1e-9)
    pf = pf_frames(4, device=dev)
    b = ppf.analyze_batch(pf, 1 / 0.390625, num_pickets=10)
    for other in (pf.to(torch.float64) - 300.0, (pf.to(torch.int32) - 32768).to(torch.int16)):
        r = ppf.analyze_batch(other, 1 / 0.390625, num_pickets=10)
        if other.dtype == torch.int16 and bool((r.status == 3).all()):
            continue                                            # (a range beyond 32767 is refused like the reference's overflow)
        assert torch.equal(r.status, b.status) and torch.equal(torch.nan_to_num(r.position, nan=-1.0), torch.nan_to_num(b.position, nan=-1.0))


@pytest.mark.gpu
def test_edge_plane32_vs_exact_path(dev):
    """Round 6: the packed-float32 edge kernel (pl_edge_plane32) and the bracketed consumers against the exact float64 path on
    random slices, ragged shapes and CatPhan slices: stored values within the bracket (the distance is measured), exact zeros,
    exact extrema, identical histogram / threshold / mask / region table; then the whole localisation with the knob on and
    off gives the same ROI table."""
    import next_row_checks as checks
    from pylinac_amd import ct
    from pylinac_amd.synthetic import catphan_volume

    worst = checks.check_edge_plane32(dev, catphan_slices=tuple(range(0, 80, 8)), sigmas=(1, 2))
    assert worst <= 32
    print("largest distance of a stored value from RN32(exact), in float32 bit patterns:", worst)
    x = torch.from_numpy(catphan_volume(4001)).to(dev)
    on = ct.phantom_roi_batch(x, 0.5)
    try:
        ct.EDGE_PLANE32 = False
        off = ct.phantom_roi_batch(x, 0.5)
    finally:
        ct.EDGE_PLANE32 = True
    assert np.array_equal(on, off, equal_nan=True)
