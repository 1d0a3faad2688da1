"""CPU (build container only: needs /root/reference): the REFERENCE'S OWN analyzer code executed on the shim.

INTEGRATION.md's claim is that pylinac's analyzers keep working when the array-level primitives they call are rebound to
``pylinac_amd``.  Here the reference is imported read-only through ``oracle/ref_loader.py`` and

  * ``PicketFence.analyze()`` (pylinac/picketfence.py:636-845) runs UNCHANGED with its image an ``pylinac_amd`` ArrayImage
    and its ``MultiProfile`` / ``FWXMProfilePhysical`` rebound to the shim's classes,
  * ``WLBaseImage._clean_edges`` / ``find_field_centroids`` (pylinac/winston_lutz.py:764-780, 1109-1133) run UNCHANGED as
    methods of a shim image,
  * ``BaseImage.compute()`` semantics hold for the reference's own ``MetricBase`` subclasses on a shim image,

with the kernels of the emulated library underneath (tests/emu_backend.py).  Results must equal the unpatched run.
"""
import os
import sys
from unittest import mock

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pylinac_oracle as oracle  # noqa: E402  (checker only)
from oracle import ref_loader  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="needs /root/reference (build container)")


def _pf_frame(h, w, pixel, seed):
    from pylinac_amd.synthetic import pf_frames

    return pf_frames(1, h, w, seed0=seed, pixel_mm=pixel, pickets=7, picket_spacing_mm=30.0, gap_mm=3.0,
                     blur_mm=2.0)[0].numpy()


def _run_reference_pf(pfm, image_cls, raw, dpmm):
    class PFImg(image_cls):                  # what PFDicomImage adds to the array image (picketfence.py:204-260)
        _central_axis = None

        def adjust_for_sag(self, sag, orientation):
            pass

    im = PFImg(raw.copy(), dpi=dpmm * 25.4, sid=1000)
    im.crop(pixels=int(round(3 * im.dpmm)))             # picketfence.py:214-215
    im.ground()
    im.normalize()                                       # picketfence.py:322-323
    pf = pfm.PicketFence(None)                           # skips image loading (picketfence.py:315)
    pf.image = im
    pf.analyze(orientation="Up-Down")                    # the reference's own analyze()
    meas = np.array([[m.leaf_num, m.picket_num, m.position[0], m._approximate_idx] for m in pf.mlc_meas])
    return meas, float(pf.max_error), float(pf.abs_median_error)


def test_reference_picketfence_analyze_runs_unchanged_on_the_shim():
    from emu_backend import emulated_device

    pfm, image = ref_loader.ref("picketfence"), ref_loader.ref("core.image")
    raw = _pf_frame(400, 520, 0.78125, 2000)
    want = _run_reference_pf(pfm, image.ArrayImage, raw, 1 / 0.78125)
    with emulated_device():
        from pylinac_amd import image as shim_image, profile as shim_profile

        with mock.patch.object(pfm, "MultiProfile", shim_profile.MultiProfile), \
                mock.patch.object(pfm, "FWXMProfilePhysical", shim_profile.FWXMProfilePhysical):
            got = _run_reference_pf(pfm, shim_image.ArrayImage, raw, 1 / 0.78125)
    assert want[0].shape == got[0].shape and want[0].shape[0] > 50
    assert np.array_equal(want[0][:, :2], got[0][:, :2]) and np.array_equal(want[0][:, 3], got[0][:, 3])
    assert np.allclose(want[0][:, 2], got[0][:, 2], rtol=0, atol=1e-9)
    assert abs(want[1] - got[1]) < 1e-9 and abs(want[2] - got[2]) < 1e-9


def test_reference_winston_lutz_methods_run_unchanged_on_a_shim_image():
    from emu_backend import emulated_device

    wl, image = ref_loader.ref("winston_lutz"), ref_loader.ref("core.image")
    g = np.load(os.path.join(ROOT, "tests", "golden", "wl.npz"))

    def run(image_cls, frame):
        class W(image_cls):
            pass

        W._clean_edges = wl.WLBaseImage._clean_edges                        # the reference's own functions
        W.find_field_centroids = wl.WLBaseImage.find_field_centroids
        img = W(frame.copy(), dpi=25.4 / 0.336)
        inverted = img.check_inversion_by_histogram(percentiles=(0.01, 50, 99.99))
        img._clean_edges()
        img.ground()
        img.normalize()
        p = img.find_field_centroids(is_open_field=False)[0]
        return inverted, img.array.shape, (p.x, p.y)

    for k in (0, 6, 7):          # plain, inverted polarity, edges cropped twice
        want = run(image.ArrayImage, g["frames"][k])
        with emulated_device():
            from pylinac_amd import image as shim_image

            got = run(shim_image.ArrayImage, g["frames"][k])
        assert want == got, k
        assert want[2] == tuple(g["record"][k, :2]), k      # and both equal the golden of the full reference run


def test_compute_contract_with_the_references_own_metric_base():
    from emu_backend import emulated_device

    mi = ref_loader.ref("metrics.image")

    class MeanOfCentralBox(mi.MetricBase):
        name = "central mean"

        def calculate(self):
            a = self.image.array
            h, w = a.shape
            return float(a[h // 2 - 2:h // 2 + 2, w // 2 - 2:w // 2 + 2].mean())

    class Vandal(mi.MetricBase):
        name = "vandal"

        def calculate(self):
            self.image.array[0, 0] += 1
            return 0

    with emulated_device():
        from pylinac_amd import image as shim_image

        img = shim_image.ArrayImage(np.arange(100, dtype=np.float64).reshape(10, 10))
        assert img.compute(MeanOfCentralBox()) == np.arange(100.0).reshape(10, 10)[3:7, 3:7].mean()
        both = img.compute([MeanOfCentralBox(), MeanOfCentralBox()])
        assert list(both) == ["central mean-1", "central mean-2"] and len(img.metrics) == 3
        assert set(img.metric_values) == {"central mean", "central mean-1", "central mean-2"}
        with pytest.raises(RuntimeError, match="modified an image"):
            img.compute(Vandal())
        assert img.center.x == 4.5 and img.center.y == 4.5


def test_find_peaks_exact_ties_vs_the_reference():
    """Where peaks tie EXACTLY on the sort key the reference's top-``max_number`` choice comes from ``np.argsort``'s
    default sort (pylinac/core/profile.py:2616-2618) -- on this numpy an AVX-dispatched, unstable sort even for a handful
    of elements -- while this backend keeps the stable order.  Enumerated here: profiles whose peaks tie on height and
    prominence, 2 .. 40 peaks, every cut.  The divergence is bounded: both keep the same multiset of key values, both
    keep EVERY peak whose key is strictly above the value at the cut, and only members of the one group tied at the cut
    may be exchanged.  Without ties at the cut the results are identical."""
    from emu_backend import emulated_device

    prof = ref_loader.ref("core.profile")
    rng = np.random.default_rng(8)
    differing = total = 0
    for npk in (2, 3, 5, 8, 12, 16, 17, 24, 40):
        heights = rng.choice([1.0, 2.0, 3.0], size=npk)            # many exact ties
        x = np.zeros(4 * npk + 1)
        x[2::4][:npk] = heights
        for key in ("prominences", "peak_heights"):
            for k in sorted({1, 2, npk // 2, npk - 1, npk}):
                if k < 1:
                    continue
                ri, rp = prof.find_peaks(x, max_number=k, peak_sort=key)
                with emulated_device():
                    from pylinac_amd import profile as shim

                    gi, gp = shim.find_peaks(x, max_number=k, peak_sort=key)
                total += 1
                assert len(gi) == len(ri) == k
                assert np.array_equal(np.sort(rp[key]), np.sort(gp[key])), (npk, key, k)   # same key values kept
                cut = np.sort(rp[key])[0]
                assert set(ri[rp[key] > cut]) == set(gi[gp[key] > cut]), (npk, key, k)     # everything above the cut
                n_tied_at_cut = int(np.sum(heights == cut))
                if np.sum(rp[key] == cut) == n_tied_at_cut:          # the whole tied group fits: no choice to make
                    assert np.array_equal(ri, gi), (npk, key, k)
                differing += int(not np.array_equal(ri, gi))
    print(f"exact-tie cases where the stable order departs from numpy's argsort: {differing} of {total}")


def test_find_peaks_distance_ties_vs_the_reference():
    """The SECOND exact-tie site: scipy's ``_select_by_peak_distance`` walks the candidates in
    ``np.argsort(priority)`` order (default kind: x86-simd-sort on this CPU, unstable), reached through
    ``peak_separation > 0`` (pylinac/core/profile.py:2605-2612).  Integer-valued / plateaued profiles -- ``np.max`` of a uint16
    image along an axis (pylinac/starshot.py:216-217), medians of integer windows -- tie exactly, and two tied candidates
    closer than ``distance`` suppress each other in whichever order the sort left them.  The device walks them in the
    STABLE order (later index first among equals, csrc/peaks_device.h distance filter).  Held here on 216 calls:
      * device == scipy's algorithm with ``argsort(kind="stable")`` (oracle ``impl="restated"``) in every call;
      * device == the live reference whenever no two tied candidates lie within ``distance`` of each other (tied
        candidates further apart commute, so the sort's order cannot matter);
      * otherwise BOTH results are valid outcomes of scipy's rule: kept peaks at least ``distance`` apart, every dropped
        candidate has a kept peak at least as high within ``distance``, and a candidate strictly higher than every
        other candidate within ``distance`` is kept by both.  (The kept heights' multiset may differ: a different winner
        of a tie suppresses different neighbours.)"""
    from emu_backend import emulated_device

    prof = ref_loader.ref("core.profile")
    rng = np.random.default_rng(11)
    differing = total = 0
    for trial in range(24):
        n = int(rng.integers(60, 400))
        if trial % 3 == 0:
            x = rng.integers(0, 6, n).astype(float)
        elif trial % 3 == 1:
            x = np.repeat(rng.integers(0, 5, n // 3 + 1), 3)[:n].astype(float)        # plateaus
        else:
            x = np.round(np.abs(rng.normal(size=n).cumsum()))
        for thr in (0, 0.1, 0.3):
            for sep in (0.02, 0.05, 0.2):
                kw = dict(threshold=thr, peak_separation=sep)
                ri, rp = prof.find_peaks(x.copy(), **kw)
                si, sp = oracle.find_peaks(x.copy(), impl="restated", **kw)
                with emulated_device():
                    from pylinac_amd import profile as shim

                    gi, gp = shim.find_peaks(x.copy(), **kw)
                total += 1
                assert np.array_equal(gi, si), (trial, kw)
                for k in sp:
                    assert np.array_equal(gp[k], sp[k]), (trial, kw, k)
                # the candidates of the distance filter: every local maximum at or above the height threshold
                ci, cp = prof.find_peaks(x.copy(), threshold=thr, peak_separation=0)
                ch = cp["peak_heights"]
                dist = max(int(sep * n), 1)
                near = np.abs(ci[:, None] - ci[None, :]) < dist
                np.fill_diagonal(near, False)
                tied_near = bool((near & (ch[:, None] == ch[None, :])).any())
                if not tied_near:
                    assert np.array_equal(gi, ri), (trial, kw)
                differing += int(not np.array_equal(gi, ri))
                dominant = ci[~(near & (ch[None, :] >= ch[:, None])).any(axis=1)]
                for kept in (ri, gi):
                    assert len(kept) < 2 or np.diff(kept).min() >= dist
                    assert set(dominant) <= set(kept)
                    kh = x[kept]
                    for c, hc in zip(ci, ch):
                        if c not in kept:
                            assert (kh[np.abs(kept - c) < dist] >= hc).any(), (trial, kw, c)
    print(f"distance-filter tie cases where the stable order departs from numpy's argsort: {differing} of {total}")


# ------------------------------------------------------------------------------------------------------------------------
# the other analyzers.  Each test runs the REFERENCE'S OWN analyzer code twice -- once on the reference's
# image / profile classes, once with those classes rebound to pylinac_amd's (emulated kernels underneath) -- and compares.
# tests/golden/make_dropin_golden.py freezes the first run so that the GPU suite can repeat the second on the MI355X.
def _class_on(module, class_name, **rebound):
    """The reference's own ``class X(Base): ...`` statement executed again with module-level names rebound (a subclass binds
    its base when the statement runs, so patching the module attribute afterwards is not enough)."""
    import inspect
    import textwrap

    ns = dict(vars(module))
    ns.update(rebound)
    exec(compile(textwrap.dedent(inspect.getsource(getattr(module, class_name))), module.__file__, "exec"), ns)
    return ns[class_name]


def _run_reference_starshot(ss, image_cls, arr, dpi, kw):
    rw = ref_loader.ref("core.warnings")
    s = object.__new__(ss.Starshot)                      # the reference only loads files: build the analyzer around an array
    rw.WarningCollectorMixin.__init__(s)
    s.image = image_cls(arr.copy(), dpi=dpi, sid=1000)
    s.wobble = ss.Wobble()
    s.tolerance = 1
    s.analyze(**kw)                                      # pylinac/starshot.py:230-301, unchanged
    cp = s.circle_profile
    return dict(profile=np.asarray(cp.values, float), circle=np.array([cp.center.x, cp.center.y, cp.radius, cp.diameter], float),
                peaks=np.array([[p.idx, p.value, p.x, p.y] for p in cp.peaks], float),
                lines=np.array([[ln.point1.x, ln.point1.y, ln.point2.x, ln.point2.y] for ln in s.lines.lines], float),
                wobble=np.array([s.wobble.center.x, s.wobble.center.y, s.wobble.radius, s.wobble.radius_mm, s.wobble.diameter_mm]),
                angles=np.array(s.angles, float), passed=bool(s.passed))


def test_reference_starshot_analyze_runs_unchanged_on_the_shim(golden):
    """``Starshot.analyze()`` (pylinac/starshot.py:230-401): inversion check, ground, FW80M start point on ``FWXMProfile``,
    ``StarProfile`` -- the reference's own subclass, re-based on the shim's ``CollapsedCircleProfile`` -- with
    ``image.dist2edge_min`` (starshot.py:794), roll / Gaussian filter / ground / ``find_fwxm_peaks``, ``LineManager`` over the
    profile's peak points, the Nelder-Mead wobble from ``circle_profile.center.as_array()``."""
    import ast

    from emu_backend import emulated_device

    ss, image = ref_loader.ref("starshot"), ref_loader.ref("core.image")
    g = golden("starshot")
    for name in ("four", "inverted", "nofwhm"):
        arr, dpi, kw = g[f"{name}.frame"], float(g[f"{name}.dpi"]), ast.literal_eval(str(g[f"{name}.kw"]))
        want = _run_reference_starshot(ss, image.ArrayImage, arr, dpi, kw)
        with emulated_device():
            from pylinac_amd import image as shim_image, profile as shim_profile

            star = _class_on(ss, "StarProfile", CollapsedCircleProfile=shim_profile.CollapsedCircleProfile)
            with mock.patch.object(ss, "StarProfile", star), mock.patch.object(ss, "FWXMProfile", shim_profile.FWXMProfile):
                got = _run_reference_starshot(ss, shim_image.ArrayImage, arr, dpi, kw)
        assert np.array_equal(want["peaks"][:, 0], got["peaks"][:, 0]) and want["passed"] == got["passed"], name
        for key in ("profile", "circle", "peaks", "lines", "angles"):
            assert np.allclose(want[key], got[key], rtol=0, atol=1e-9), (name, key)
        assert np.allclose(want["wobble"], got["wobble"], rtol=0, atol=1e-6), name           # (Nelder-Mead, fatol 1e-3)
        assert np.allclose(want["wobble"], g[f"{name}.wobble"], rtol=0, atol=1e-9), name     # = the frozen reference run


def _run_reference_field_analysis(fa, image_cls, arr, dpi, kw):
    f = object.__new__(fa.FieldAnalysis)                 # FieldAnalysis.__init__ (field_analysis.py:448-470) minus image.load
    f._path = "array"
    f.image = image_cls(arr.copy(), dpi=dpi)
    f._is_analyzed = False
    f._from_device = False
    f.image.check_inversion_by_histogram()
    f.analyze(**kw)                                      # field_analysis.py:562-965, unchanged

    def flat(d):
        out = {}
        for k, v in d.items():
            out[k] = np.asarray(v, dtype=float).reshape(-1)
        return out

    res = flat(f._results)
    res.update({"protocol." + k: v for k, v in flat(f._extra_results).items()})
    res["horiz.values"], res["vert.values"] = np.asarray(f.horiz_profile.values, float), np.asarray(f.vert_profile.values, float)
    res["horiz.x"], res["vert.x"] = np.asarray(f.horiz_profile.x_indices, float), np.asarray(f.vert_profile.x_indices, float)
    return res


def test_reference_field_analysis_analyze_runs_unchanged_on_the_shim(golden):
    """``FieldAnalysis.analyze()`` (pylinac/field_analysis.py:562-965): centre search on ``SingleProfile(np.sum(image, axis))``,
    the two strip profiles, penumbrae, field sizes, CAX / beam-centre distances, "top" positions, slopes and the protocol's
    flatness / symmetry functions over ``SingleProfile.field_data`` -- with ``SingleProfile`` rebound to the shim's and the
    image a shim ``ArrayImage``.  Every number of ``_results`` / ``_extra_results`` and both processed profiles must agree."""
    from emu_backend import emulated_device

    fa, image = ref_loader.ref("field_analysis"), ref_loader.ref("core.image")
    arr = golden("field_strips")["frames"][1]
    cases = [dict(protocol=fa.Protocol.VARIAN),
             dict(protocol=fa.Protocol.ELEKTA, edge_detection_method="FWHM", centering="Geometric center", vert_width=0.05, horiz_width=0.05),
             dict(protocol=fa.Protocol.SIEMENS, edge_detection_method="Inflection Hill", is_FFF=True, interpolation="Spline",
                  normalization_method="Max", hill_window_ratio=0.1),
             dict(protocol=fa.Protocol.NONE, centering="Manual", vert_position=0.45, horiz_position=0.55, interpolation=None)]
    for n, kw in enumerate(cases):
        want = _run_reference_field_analysis(fa, image.ArrayImage, arr, 100, kw)
        with emulated_device():
            from pylinac_amd import image as shim_image, profile as shim_profile

            with mock.patch.object(fa, "SingleProfile", shim_profile.SingleProfile):
                got = _run_reference_field_analysis(fa, shim_image.ArrayImage, arr, 100, kw)
        assert set(want) == set(got), n
        hill = kw.get("edge_detection_method") == "Inflection Hill"
        for key in want:
            # downstream of a Hill fit: MINPACK stops at 1.5e-8 relative and numpy's vectorised pow is not bit-reproducible
            tol = dict(rtol=1e-5, atol=1e-5) if hill else dict(rtol=1e-9, atol=1e-9)
            assert want[key].shape == got[key].shape and np.allclose(want[key], got[key], equal_nan=True, **tol), (n, key, want[key], got[key])


def _run_reference_ctp528(ct, image_cls, vol, fit_zx, fit_zy, mmpp, s):
    stack = [image_cls(sl.copy()) for sl in vol]
    m = object.__new__(ct.CTP528CP504)                   # CatPhanModule.__init__ wants a whole CatPhan: set what the module reads
    m.origin_slice, m._offset, m.slice_spacing = int(s), 0, 2.5
    m._phantom_center_func = (np.poly1d(fit_zx), np.poly1d(fit_zy))
    m.scaling_factor, m.mm_per_pixel, m.catphan_roll, m.roi_size_factor = 1, mmpp, 0.0, 1
    m.image = image_cls(ct.combine_surrounding_slices(stack, int(s), slices_plusminus=3, mode="max"))
    prof = m.circle_profile                              # pylinac/ct.py:1562-1580, unchanged: the image OBJECT is the array
    mtf = m.mtf                                          # pylinac/ct.py:1511-1544, unchanged
    return dict(profile=np.asarray(prof.values, float), maxs=np.asarray(mtf.maximums, float), mins=np.asarray(mtf.minimums, float),
                rmtf=np.asarray(list(mtf.norm_mtfs.values()), float), circle=np.array([prof.center.x, prof.center.y, prof.radius]),
                mtf50=float(mtf.relative_resolution(50)))


def test_reference_ctp528_module_runs_unchanged_on_the_shim(golden):
    """``CTP528CP504.circle_profile`` / ``.mtf`` (pylinac/ct.py:1511-1580) on a synthetic CatPhan volume: the +-3-slice maximum
    image, ``CollapsedCircleProfile(phan_center, radius, image_array=<the image object>, ...)``, ``filter(0.001, "gaussian")``,
    ``ground()``, per line-pair region ``find_peaks`` / ``find_valleys`` and the reference's ``MTF`` class on their values --
    with ``CollapsedCircleProfile`` rebound to the shim's and the slice images shim ``ArrayImage`` objects."""
    from emu_backend import emulated_device

    ct, image = ref_loader.ref("ct"), ref_loader.ref("core.image")
    g = golden("ctp528")
    vol, mmpp = g["volume"], float(g["mmpp"])
    for s in (int(g["resolution_slice"]), int(g["slices"][0])):
        k = int(np.flatnonzero(g["slices"] == s)[0])
        if int(g["nregions"][k]) == 0:
            continue
        want = _run_reference_ctp528(ct, image.ArrayImage, vol, g["fit_zx"], g["fit_zy"], mmpp, s)
        with emulated_device():
            from pylinac_amd import image as shim_image, profile as shim_profile

            with mock.patch.object(ct, "CollapsedCircleProfile", shim_profile.CollapsedCircleProfile):
                got = _run_reference_ctp528(ct, shim_image.ArrayImage, vol, g["fit_zx"], g["fit_zy"], mmpp, s)
        for key in want:
            assert np.allclose(want[key], got[key], rtol=1e-9, atol=1e-9), (s, key)
        assert np.allclose(want["profile"], g["profiles"][k], rtol=0, atol=1e-9)          # = the frozen reference run
        assert np.allclose(want["rmtf"], g["rmtf"][k][: len(want["rmtf"])], rtol=1e-9, atol=1e-9)


def test_reference_wl_analyze_whole_method_runs_unchanged_on_the_shim(golden):
    """``WLBaseImage.analyze()`` -- the WHOLE method (pylinac/winston_lutz.py:669-762): inversion check, ``_clean_edges``,
    ground, normalize, ``find_field_centroids``, ``find_field_matches`` / ``find_bb_matches`` against the projected nominal BB
    position, ``find_bb_centroids`` = ``self.compute(SizedDiskLocator.from_center_physical(...))``, the optional shift vector,
    ``BBFieldMatch`` records -- as methods of a shim ``ArrayImage`` with ``SizedDiskLocator`` bound to
    ``pylinac_amd.metrics.SizedDiskLocator``.  The unpatched run needs scikit-image (the reference's own ``find_features``), so
    it was made under python3.9 by tests/golden/skimage_dropin_wl_py39.py (same frames, same class recipe) and frozen in
    tests/golden/dropin_wl.npz; the points the two packages' ``Point`` classes exchange here (reference field points against
    shim BB points in ``distance_to`` / ``-`` / ``BBFieldMatch``) are the mixed use an integration produces."""
    import importlib.util

    from emu_backend import emulated_device

    spec = importlib.util.spec_from_file_location("dropin_wl_recipe", os.path.join(ROOT, "tests", "golden", "skimage_dropin_wl_py39.py"))
    wl, image, geometry = ref_loader.ref("winston_lutz"), ref_loader.ref("core.image"), ref_loader.ref("core.geometry")
    src = open(spec.origin).read()
    recipe = {"wl": wl, "np": np}
    # the recipe functions only (class construction + run); the module's own imports / __main__ part are for python3.9
    exec(compile(src[src.index("BORROWED = "):src.index('if __name__ == "__main__":')], spec.origin, "exec"), recipe)
    g, frames = golden("dropin_wl"), golden("wl")
    pixel = float(frames["pixel_mm"])
    kws = {0: {}, 6: {}, 7: {}, 8: dict(bb_proximity_mm=30), 2: dict(shift_vector=geometry.Vector(x=0.4, y=-0.3, z=0.2), snap_tolerance=1)}
    with emulated_device():
        from pylinac_amd import image as shim_image, metrics as shim_metrics

        W = recipe["wl_image_class"](wl, image, shim_image.ArrayImage)
        with mock.patch.object(wl, "SizedDiskLocator", shim_metrics.SizedDiskLocator):
            for row, k, gantry, couch in zip(g["record"], g["frame_index"], g["gantry"], g["couch"]):
                got = recipe["run"](W, frames["frames"][int(k)], pixel, float(gantry), float(couch), **kws[int(k)])
                assert np.array_equal(got[[0, 1, 10, 11]], row[[0, 1, 10, 11]]), k        # field CAX, cleaned shape: exact
                assert np.allclose(got, row, rtol=0, atol=1e-9), (k, got - row)


# what the hot path leaves to the host package: drawing, file metadata, file loading (DESIGN.md "Out of scope")
_OUT_OF_SCOPE = {"plot", "plotly", "plot2axes", "plot_gamma", "plot_metrics", "as_dicom", "date_created", "from_multiples",
                 "truncated_path"}


def test_public_members_of_the_image_and_profile_classes_vs_the_reference():
    """``dir()`` diff of every profile class, the profile-module functions and the array image class against the live reference:
    nothing public is missing except plotting / file members.  (VERDICT r4 "missing" 2.)"""
    import inspect

    rp, ri = ref_loader.ref("core.profile"), ref_loader.ref("core.image")
    from pylinac_amd import image as si, profile as sp

    def public(c):
        return {n for n in dir(c) if not n.startswith("_")}

    missing = {}
    for name, cls in inspect.getmembers(rp, inspect.isclass):
        if cls.__module__ != rp.__name__:
            continue
        shim = getattr(sp, name, None)
        assert shim is not None, f"profile.{name} missing"
        gap = public(cls) - public(shim) - _OUT_OF_SCOPE
        if gap:
            missing[name] = sorted(gap)
    for name, fn in inspect.getmembers(rp, inspect.isfunction):
        if fn.__module__ == rp.__name__ and not name.startswith("_"):
            assert hasattr(sp, name), f"profile.{name} missing"
    for name in ("BaseImage", "ArrayImage"):
        gap = public(getattr(ri, name)) - public(getattr(si, name)) - _OUT_OF_SCOPE
        gap -= {"path", "base_path", "source", "metadata"} & gap        # (attributes of file-backed images)
        if gap:
            missing[name] = sorted(gap)
    assert not missing, missing
    # and the constructors take the reference's parameters, in order, with its defaults
    for name in ("FWXMProfile", "FWXMProfilePhysical", "InflectionDerivativeProfile", "HillProfile", "SingleProfile", "MultiProfile",
                 "CircleProfile", "CollapsedCircleProfile", "ProfileBase"):
        want = inspect.signature(getattr(rp, name).__init__).parameters
        got = inspect.signature(getattr(sp, name).__init__).parameters
        assert list(want) == list(got), (name, list(want), list(got))
        for k in want:
            if want[k].default is not inspect.Parameter.empty and not isinstance(want[k].default, (np.ndarray,)):
                a, b = want[k].default, got[k].default
                assert (getattr(a, "value", a) == getattr(b, "value", b)), (name, k, a, b)


def test_profile_mixin_members_behave_like_the_references():
    """ProfileMixin (pylinac/core/profile.py:86-153) on every 1-D profile class, ``x_at_x``, ``SingleProfile.resample`` /
    ``gamma``, ``compute`` with the reference's OWN ``ProfileMetric`` subclasses (pylinac/metrics/profile.py), and the caching
    rules: ``center_idx`` & co. are fixed at first use while ``field_edge_idx`` follows the current values."""
    import warnings

    from emu_backend import emulated_device

    rp, rm = ref_loader.ref("core.profile"), ref_loader.ref("metrics.profile")
    x = np.arange(160, dtype=float)
    v = 80.0 / (1 + np.exp(-(x - 40) / 3.0)) / (1 + np.exp((x - 118) / 3.5)) + 4 + 0.05 * np.sin(x)
    u16 = (v * 300).astype(np.uint16)

    def exercise(mod, img_mod=None):
        out = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for cname, kw in (("FWXMProfile", {}), ("InflectionDerivativeProfile", {}), ("HillProfile", {}),
                              ("FWXMProfilePhysical", dict(dpmm=2.5))):
                p = getattr(mod, cname)(v.copy(), **kw)
                first = (p.center_idx, p.field_width_px, p.geometric_center_idx, p.cax_index)
                p.invert()
                out[cname + ".invert"] = np.asarray(p.values, float)
                p.stretch(min=1, max=7)
                out[cname + ".stretch"] = np.asarray(p.values, float)
                p.filter(size=5, kind="median")
                mn = p.ground()
                p.normalize()
                out[cname + ".chain"] = np.asarray(p.values, float)
                out[cname + ".ground_min"] = np.array([mn])
                # cached at first use (cached_property in the reference) although the values were inverted since
                out[cname + ".sticky"] = np.array([p.center_idx, p.field_width_px, p.geometric_center_idx, p.cax_index]) - np.array(first)
                out[cname + ".x_at_x"] = np.array([p.x_at_x(10.25), p.x_at_x_idx(10.25)])
                q = getattr(mod, cname)(u16.copy(), **kw)
                q.bit_invert()
                out[cname + ".bit_invert"] = np.asarray(q.values, float)
                q.convert_to_dtype(np.uint8)
                out[cname + ".convert"] = np.asarray(q.values, float)
                assert q.values.dtype == np.uint8 and len(q) == 160 and q[3] == q.values[3]
            f = mod.FWXMProfile(v.copy())
            e0 = f.field_edge_idx("left")
            f.filter(size=0.08, kind="gaussian")
            out["edge_follows_values"] = np.array([e0, f.field_edge_idx("left")])      # searched on the CURRENT values
            m = mod.MultiProfile(v.copy())
            m.invert()
            m.stretch()
            out["multi"] = np.asarray(m.values, float)
            s = mod.SingleProfile(v.copy(), dpmm=2.0, interpolation_resolution_mm=0.2)
            s.invert()
            s.stretch(0, 2)
            out["single.mixin"] = np.asarray(s.values, float)
            out["single.fwxm_after"] = np.array([s.fwxm_data(50)["width (exact)"]])     # peaks of the NEW values, look-ups of the OLD
            s2 = mod.SingleProfile(v.copy(), dpmm=2.0, interpolation_resolution_mm=0.2)
            r = s2.resample(interpolation_resolution_mm=0.1)
            out["single.resample"] = np.asarray(r.values, float)
            out["single.resample_x"] = np.asarray(r.x_indices, float)
            out["single.gamma"] = s2.gamma(mod.SingleProfile(v * 1.01, dpmm=2.0, interpolation_resolution_mm=0.2), distance_to_agreement=1,
                                           dose_to_agreement=1)
            # the reference's own profile metrics through ProfileBase.compute
            pm = mod.FWXMProfilePhysical(v.copy(), dpmm=2.0)
            vals = pm.compute(metrics=[rm.PenumbraLeftMetric(), rm.PenumbraRightMetric(), rm.FlatnessDifferenceMetric(),
                                       rm.SymmetryPointDifferenceMetric(), rm.CAXToLeftEdgeMetric(), rm.TopDistanceMetric()])
            out["metrics"] = np.array([float(vals[k]) for k in sorted(vals)])
            one = mod.FWXMProfilePhysical(v.copy(), dpmm=2.0).compute(metrics=rm.FlatnessRatioMetric())
            out["metric_single"] = np.array([float(one)])
            assert sorted(pm.metric_values) == sorted(vals) and len(pm.metrics) == 6
            out["stretch_fn"] = np.asarray(mod.stretch(u16.copy(), fill_dtype=np.uint8), float)
        return out

    want = exercise(rp)
    with emulated_device():
        from pylinac_amd import profile as sp

        got = exercise(sp)
    assert set(want) == set(got)
    for k in want:
        assert want[k].shape == got[k].shape and np.allclose(want[k], got[k], rtol=1e-9, atol=1e-9, equal_nan=True), k
    assert not want["FWXMProfile.sticky"].any() and want["edge_follows_values"][0] != want["edge_follows_values"][1]


def test_image_members_the_analyzers_call():
    """BaseImage.dist2edge_min / physical_shape / flat (pylinac/core/image.py:536, 817, 1095) and the geometry value types the
    image and profile classes hand out, against the reference's own."""
    from emu_backend import emulated_device

    ri, rg = ref_loader.ref("core.image"), ref_loader.ref("core.geometry")
    a = np.arange(12 * 17, dtype=np.float64).reshape(12, 17)
    ref = ri.ArrayImage(a.copy(), dpi=50.8)
    with emulated_device():
        from pylinac_amd import geometry as sg, image as si

        img = si.ArrayImage(a.copy(), dpi=50.8)
        assert isinstance(img, si.BaseImage)
        for pt in ((3.2, 4.9), (16.0, 0.5), (8.5, 11.75)):
            assert img.dist2edge_min(pt) == ref.dist2edge_min(pt) == ref.dist2edge_min(rg.Point(pt)) == img.dist2edge_min(sg.Point(pt))
            assert img.dist2edge_min(rg.Point(pt)) == ref.dist2edge_min(sg.Point(pt))          # either package's points
        assert img.physical_shape == ref.physical_shape and list(img.flat) == list(ref.flat)
        assert (img.center.x, img.center.y) == (ref.center.x, ref.center.y)
        # points: construction, copy constructor across packages, arithmetic, distances
        p, q = sg.Point(1.5, -2, 4, idx=3, value=9.0), rg.Point(1.5, -2, 4, idx=3, value=9.0)
        assert list(p.as_array()) == list(q.as_array()) and p.dict() == q.dict() and repr(p) == repr(q)
        assert p == sg.Point(q) and q == rg.Point(p) and sg.Point((1, 2)).z == 0 == rg.Point((1, 2)).z
        assert p.distance_to(rg.Point(4, 2, 4)) == q.distance_to(sg.Point(4, 2, 4)) == 5 * 0 + q.distance_to(rg.Point(4, 2, 4))
        c = sg.Circle((3, 4), 2.5)
        rc = rg.Circle((3, 4), 2.5)
        assert (c.area, c.diameter, c.as_dict()) == (rc.area, rc.diameter, rc.as_dict())
        assert p.distance_to(rc) == p.distance_to(c) == q.distance_to(rc)
        d, rd = p - sg.Point(0.5, 1, 1), q - rg.Point(0.5, 1, 1)
        assert (d.x, d.y, d.z, (d / 2).x, d.as_scalar()) == (rd.x, rd.y, rd.z, (rd / 2).x, rd.as_scalar())
        assert (sg.Point(2, 3) * 4).x == (rg.Point(2, 3) * 4).x == 8 and sg.Point(2.6, 3.4, as_int=True).x == 3
