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


def _decision_frames():
    rng = np.random.default_rng(7)
    out = []
    for k in range(6):
        h, w = (180, 240) if k % 2 == 0 else (210, 160)
        yy, xx = np.mgrid[0:h, 0:w].astype(float)
        img = 2000 + 30000 * np.exp(-(((yy - h / 2) / (h / 4)) ** 4 + ((xx - w / 2) / (w / 4)) ** 4)) + rng.normal(0, 300, (h, w))
        if k in (2, 3):      # picket-like stripes along one axis
            img += 15000 * (np.sin((xx if k == 2 else yy) / 6.0) > 0.6)
        if k == 4:           # inverted
            img = 40000 - img
        img = np.clip(img, 0, 65535)
        if k in (1, 5):      # dead / hot pixels
            img.ravel()[rng.integers(0, img.size, 4)] = [0, 65535, 65535, 0]
        out.append(img.astype(np.uint16))
    return out


def test_decision_restatements_live():
    """a6: oracle.has_noise / pf_orientation / corners_inverted / clean_edges against the reference's own methods
    (PFDicomImage._has_noise, PicketFence.orientation, BaseImage.check_inversion, WLBaseImage._clean_edges), driven
    with stand-in objects that carry only the attributes those methods read."""
    pf, image, wl = ref_loader.ref("picketfence"), ref_loader.ref("core.image"), ref_loader.ref("winston_lutz")

    class Obj:
        pass

    for a in _decision_frames():
        f = Obj()
        f.array = a
        assert bool(pf.PFDicomImage._has_noise(f)) == o.has_noise(a)
        p = Obj()
        p._orientation, p.image = None, Obj()
        p.image.array = a
        assert pf.PicketFence.orientation.func(p).value == o.pf_orientation(a)
        im = image.ArrayImage(a.copy())
        im.check_inversion()
        assert (not np.array_equal(im.array, a)) == o.corners_inverted(a)
        b = a.copy()
        b[0, :40] = 65535
        b[:, -1] = 0
        b[-2:, 10:30] = 65535
        w = image.ArrayImage(b.copy())
        wl.WLBaseImage._clean_edges(w)
        assert np.array_equal(w.array, o.clean_edges(b)) and w.array.shape != b.shape
