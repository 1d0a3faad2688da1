"""CPU, build container only: drives the reference's OWN modules (read-only from /root/reference
through oracle/ref_loader.py) on fresh random inputs and checks the oracle against them.
Skipped where /root/reference does not exist (the GPU box)."""
import numpy as np
import pytest

from oracle import pylinac_oracle as o
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="/root/reference not present")


def test_filters_and_mutators_live():
    au = ref_loader.ref("core.array_utils")
    rng = np.random.default_rng(123)
    for shape in [(40, 50), (33, 17)]:
        a = rng.integers(0, 65536, shape, dtype=np.uint16)
        for size, kind in [(3, "median"), (0.1, "median"), (2, "gaussian"), (5, "gaussian")]:
            assert np.array_equal(au.filter(a, size, kind), o.filter(a, size, kind))
            s = o.resolve_filter_size(a, size)
            rest = o.gaussian_filter_restated(a, s) if kind == "gaussian" else o.median_filter_restated(a, s)
            assert np.array_equal(au.filter(a, size, kind), rest)
        for fn in ("ground", "normalize", "invert"):
            assert np.array_equal(getattr(au, fn)(a), getattr(o, fn)(a))
        assert np.array_equal(au.stretch(a.astype(float), 0, 1), o.stretch(a.astype(float), 0, 1))


def test_find_peaks_live():
    prof = ref_loader.ref("core.profile")
    rng = np.random.default_rng(5)
    for trial in range(40):
        x = np.abs(rng.normal(size=int(rng.integers(20, 500))).cumsum())
        for kw in (dict(), dict(threshold=0.3, peak_separation=0.05), dict(fwxm_height=0.3, max_number=1),
                   dict(search_region=(0.2, 0.8), max_number=2)):
            i1, p1 = prof.find_peaks(x.copy(), **kw)
            for impl in ("scipy", "restated"):
                i2, p2 = o.find_peaks(x.copy(), impl=impl, **kw)
                assert np.array_equal(i1, i2)
                for k in p1:
                    assert np.array_equal(p1[k], p2[k]), (impl, k)


def test_reference_kats_through_loader():
    """SURVEY.md Appendix C checks."""
    prof = ref_loader.ref("core.profile")
    image = ref_loader.ref("core.image")
    p = prof.FWXMProfile(np.array([0, 1, 2, 3, 4, 3, 2, 1, 0.0]))
    assert (p.field_edge_idx("left"), p.field_edge_idx("right"), p.center_idx) == (2.0, 6.0, 4.0)
    im = image.ArrayImage(np.arange(42).reshape(6, 7))
    im.filter(3)
    assert im.array[0, 0] == 1
