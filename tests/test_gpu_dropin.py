"""GPU (-m gpu): the analyzers' call sequences REPLAYED on pylinac_amd's host classes over the real HIP kernels.

tests/test_dropin_reference.py runs the reference's own ``Starshot.analyze`` / ``WLBaseImage.analyze`` / ``CTP528CP504`` /
``FieldAnalysis.analyze`` code over these classes, but only where ``/root/reference`` exists (the build container, emulated
kernels).  The GPU box has no reference tree, so here the SAME calls those methods make on the image / profile / metric objects
are written out call by call (each block cites the reference lines it replays) and the results are compared with what the
reference's own code returned on the same inputs: tests/golden/starshot.npz (make_starshot_golden.py), dropin_wl.npz
(skimage_dropin_wl_py39.py: the whole ``WLBaseImage.analyze``), ctp528.npz (skimage_ctp528_py39.py).  Nothing below calls the
batched pipelines (``pylinac_amd.starshot`` / ``winston_lutz.analyze_batch`` / ``ct.ctp528_batch``): this file is about the
per-image CLASS API an unchanged analyzer would use.  ``FieldAnalysis`` reduces to ``SingleProfile`` calls, which
``test_gpu_parity.py`` pins to the reference's own ``SingleProfile`` (single_profile.npz, profile_batch.npz, hill.npz).
"""
import ast
import copy

import numpy as np
import pytest
from scipy import ndimage, optimize

pytestmark = pytest.mark.gpu


def test_starshot_call_sequence_on_the_class_api(golden, dev):
    """Starshot.analyze -> _get_reasonable_start_point -> StarProfile (pylinac/starshot.py:197-227, 283-301, 774-811), first
    pass of ``_get_reasonable_wobble`` (the golden cases converge on it), Nelder-Mead wobble (:385-401)."""
    from pylinac_amd.geometry import Point
    from pylinac_amd.image import ArrayImage
    from pylinac_amd.profile import CollapsedCircleProfile, FWXMProfile

    g = golden("starshot")
    for name in ("four", "six_off", "inverted", "nofwhm", "float"):
        kw = ast.literal_eval(str(g[f"{name}.kw"]))
        img = ArrayImage(g[f"{name}.frame"].copy(), dpi=float(g[f"{name}.dpi"]), sid=1000)
        img.check_inversion_by_histogram(percentiles=[4, 50, 96])                       # starshot.py:283
        img.ground()
        a = img.array
        t3, l3 = int(a.shape[0] / 3), int(a.shape[1] / 3)                               # :207-211 the central third
        central = a[t3:int(t3 * 2), l3:int(l3 * 2)]
        x = round(FWXMProfile(values=np.max(central, 0), fwxm_height=80).center_idx) + l3      # :216-224
        y = round(FWXMProfile(values=np.max(central, 1), fwxm_height=80).center_idx) + t3
        start, local_max = Point(x, y), np.percentile(central, 90)
        radius, height = kw.get("radius", 0.85), kw.get("min_peak_height", 0.25) * local_max
        prof = CollapsedCircleProfile(center=start, radius=img.dist2edge_min(start) * radius, image_array=img.array,
                                      width_ratio=0.1, sampling_ratio=3)                 # StarProfile.__init__ :777-786
        prof.roll(np.where(prof.values == prof.values.min())[0][0])                      # :801-806
        prof.filter(size=0.003, kind="gaussian")
        prof.ground()
        if kw.get("fwhm", True):
            prof.find_fwxm_peaks(threshold=height, min_distance=0.02)
        else:
            prof.find_peaks(height, 0.02)
        assert np.allclose([prof.center.x, prof.center.y, prof.radius], g[f"{name}.circle"], rtol=0, atol=1e-9), name
        assert np.allclose(np.asarray(prof.values, float), g[f"{name}.profile"], rtol=0, atol=1e-9), name
        peaks = np.array([[p.idx, p.value, p.x, p.y] for p in prof.peaks], float)
        assert np.array_equal(peaks[:, 0], g[f"{name}.peaks"][:, 0]) and np.allclose(peaks, g[f"{name}.peaks"], rtol=0, atol=1e-9), name
        # LineManager.match_points + _find_wobble_minimize (:750-760, 385-401) over the profile's own peak points
        n = len(prof.peaks) // 2
        pairs = [(prof.peaks[i], prof.peaks[i + n]) for i in range(n)]

        def line_distance(p, a, b):                                                     # geometry.Line.distance_to
            lp1, lp2, pt = a.as_array(), b.as_array(), np.array([p[0], p[1], p[2] if len(p) > 2 else 0.0])
            return np.sqrt(np.sum(np.power(np.cross(lp2 - lp1, lp1 - pt), 2))) / np.sqrt(np.sum(np.power(lp2 - lp1, 2)))

        sp = copy.copy(prof.center)
        res = optimize.minimize(lambda p: max(line_distance(p, a, b) for a, b in pairs), sp.as_array(), method="Nelder-Mead",
                                options={"fatol": 0.001})
        assert np.allclose([res.x[0], res.x[1], res.fun], g[f"{name}.wobble"][:3], rtol=0, atol=1e-6), name


def test_wl_analyze_call_sequence_on_the_class_api(golden, dev):
    """WLBaseImage.analyze (pylinac/winston_lutz.py:709-725) as calls on an ArrayImage: inversion check, ``_clean_edges``
    (:1109-1133), ground, normalize, ``find_field_centroids`` (:764-780), ``find_bb_centroids`` (:788-806) through
    ``image.compute(SizedDiskLocator.from_center_physical(...))``."""
    from pylinac_amd.image import ArrayImage
    from pylinac_amd.metrics import SizedDiskLocator

    g, frames = golden("dropin_wl"), golden("wl")
    pixel = float(frames["pixel_mm"])
    for row, k in zip(g["record"], g["frame_index"]):
        if int(k) == 2:
            continue                                    # (the shift-vector case moves the BB on the host afterwards)
        img = ArrayImage(frames["frames"][int(k)].copy(), dpi=25.4 / pixel)
        img.check_inversion_by_histogram(percentiles=(0.01, 50, 99.99))
        safety_stop = np.min(img.shape) / 10                                            # _clean_edges(window_size=2)
        while safety_stop > 0:
            near_min, near_max = np.percentile(img.array, [5, 99.5])
            rng = near_max - near_min
            edge = np.concatenate([img[:2, :].flatten(), img[:, :2].flatten(), img[-2:, :].flatten(), img[:, -2:].flatten()])
            if not (edge.min() < near_min - rng / 10 or edge.max() > near_max + rng / 10):
                break
            img.crop(2)
            safety_stop -= 1
        assert (img.shape[0], img.shape[1]) == (int(row[10]), int(row[11])), k
        img.ground()
        img.normalize()
        lo, hi = np.percentile(img.array, [5, 99.9])
        filled = ndimage.binary_fill_holes(img.as_binary((hi - lo) / 2 + lo))           # the reference's own scipy calls
        cy, cx = ndimage.center_of_mass(filled)
        assert (cx, cy) == (row[0], row[1]), k
        tol = float(np.interp(5.0, (1.5, 30), (2, 4)))                                  # _calculate_bb_tolerance
        pts = img.compute(metrics=SizedDiskLocator.from_center_physical(
            expected_position_mm=(0, 0), search_window_mm=(45.0, 45.0), radius_mm=2.5, radius_tolerance_mm=tol, invert=True,
            name="BB"))
        assert len(pts) == 1 and abs(pts[0].x - row[2]) < 1e-9 and abs(pts[0].y - row[3]) < 1e-9, k
        assert "BB" in img.metric_values


def test_ctp528_call_sequence_on_the_class_api(golden, dev):
    """CTP528CP504.circle_profile / .mtf (pylinac/ct.py:1511-1580): CollapsedCircleProfile over the IMAGE OBJECT of the +-3-slice
    maximum, Gaussian filter, ground, per-region find_peaks / find_valleys."""
    from pylinac_amd.ct import CTP528_REGIONS
    from pylinac_amd.image import ArrayImage
    from pylinac_amd.profile import CollapsedCircleProfile

    g = golden("ctp528")
    vol, mmpp = g["volume"], float(g["mmpp"])
    zx, zy = np.poly1d(g["fit_zx"]), np.poly1d(g["fit_zy"])
    for k in (0, len(g["slices"]) // 2):
        s = int(g["slices"][k])
        image = ArrayImage(np.max(vol[s - 3:s + 4], axis=0))                           # combine_surrounding_slices(+-3, "max")
        prof = CollapsedCircleProfile((float(zx(s)), float(zy(s))), 47 / mmpp, image_array=image, start_angle=np.pi,
                                      width_ratio=0.04, sampling_ratio=2, ccw=True)
        prof.filter(0.001, kind="gaussian")
        prof.ground()
        assert np.allclose(np.asarray(prof.values, float), g["profiles"][k], rtol=0, atol=1e-9), s
        maxs, mins = [], []
        for start, end, npk, nval, spacing, _ in CTP528_REGIONS:                       # roi_settings, ct.py:1417-1503
            idx, vals = prof.find_peaks(min_distance=spacing, max_number=npk, search_region=(start, end))
            if len(vals) != npk:
                break
            maxs.append(vals.mean())
            _, vv = prof.find_valleys(min_distance=spacing, max_number=nval, search_region=(min(idx), max(idx)))
            mins.append(vv.mean())
        n = int(g["nregions"][k])
        assert len(maxs) == n
        assert np.allclose(maxs, g["maxs"][k][:n], rtol=1e-9, atol=1e-9) and np.allclose(mins, g["mins"][k][:n], rtol=1e-9, atol=1e-9)


def test_field_analysis_call_sequence_on_the_class_api(golden, dev):
    """FieldAnalysis.analyze -> _extract_profiles -> the result assembly (pylinac/field_analysis.py:488-561, 720-862): inversion
    check, the centre search on SingleProfile(np.sum(image, axis)), the two strip profiles, their SingleProfile objects, and
    every entry of ``_results`` / ``_extra_results`` read off ``penumbra`` / ``geometric_center`` / ``beam_center`` /
    ``field_data`` and the protocol's flatness / symmetry functions -- against the reference's own analyze() on the same frame
    (tests/golden/make_dropin_field_golden.py), four protocol / centering / edge-method / interpolation combinations."""
    import json

    import torch

    from pylinac_amd import field_analysis as pfa
    from pylinac_amd.image import ArrayImage
    from pylinac_amd.profile import SingleProfile

    g = golden("dropin_field")
    protocols = {"VARIAN": dict(symmetry=pfa.symmetry_point_difference, flatness=pfa.flatness_dose_difference),
                 "ELEKTA": dict(symmetry=pfa.symmetry_pdq_iec, flatness=pfa.flatness_dose_ratio),
                 "SIEMENS": dict(symmetry=pfa.symmetry_area, flatness=pfa.flatness_dose_difference), "NONE": {}}
    for n, kw in enumerate(json.loads(str(g["cases"]))):
        img = ArrayImage(g["frame"].copy(), dpi=float(g["dpi"]))
        img.check_inversion_by_histogram()                                              # FieldAnalysis.__init__ :470
        frame = torch.from_numpy(np.ascontiguousarray(img.array)).to(dev)[None]
        centering = kw.get("centering", "Beam center")
        vpos, hpos = kw.get("vert_position", 0.5), kw.get("horiz_position", 0.5)
        if centering != "Manual":
            vpos, hpos = pfa.determine_center(frame, centering)                        # :527-528
        hv, upper, lower = pfa.horiz_values(frame, hpos, kw.get("horiz_width", 0))      # :530-532
        vv, left, right = pfa.vert_values(frame, vpos, kw.get("vert_width", 0))
        edge = kw.get("edge_detection_method", "Inflection Derivative")
        common = dict(dpmm=img.dpmm, interpolation=kw.get("interpolation", "Linear"), interpolation_resolution_mm=0.1, ground=True,
                      edge_detection_method=edge, normalization_method=kw.get("normalization_method", "Beam center"),
                      edge_smoothing_ratio=0.003, hill_window_ratio=kw.get("hill_window_ratio", 0.15))
        hp, vp = SingleProfile(hv[0].cpu().numpy(), **common), SingleProfile(vv[0].cpu().numpy(), **common)   # :535-560
        tol = dict(rtol=1e-5, atol=1e-5) if edge == "Inflection Hill" else dict(rtol=1e-9, atol=1e-9)
        assert np.allclose(np.asarray(hp.values, float), g[f"{n}.horiz"], **tol) and np.allclose(np.asarray(vp.values, float), g[f"{n}.vert"], **tol), n
        ser, ifr = 0.2, 0.8
        v_pen, h_pen = vp.penumbra(20, 80), hp.penumbra(20, 80)                        # :768-...
        res = {"top_penumbra_mm": v_pen["left penumbra width (exact) mm"], "bottom_penumbra_mm": v_pen["right penumbra width (exact) mm"],
               "left_penumbra_mm": h_pen["left penumbra width (exact) mm"], "right_penumbra_mm": h_pen["right penumbra width (exact) mm"]}
        if edge == "Inflection Hill":
            res.update(top_penumbra_percent_mm=abs(v_pen["left gradient (exact) %/mm"]), bottom_penumbra_percent_mm=abs(v_pen["right gradient (exact) %/mm"]),
                       left_penumbra_percent_mm=abs(h_pen["left gradient (exact) %/mm"]), right_penumbra_percent_mm=abs(h_pen["right gradient (exact) %/mm"]))
        res["geometric_center_index_x_y"] = (hp.geometric_center()["index (exact)"], vp.geometric_center()["index (exact)"])
        res["beam_center_index_x_y"] = (hp.beam_center()["index (exact)"], vp.beam_center()["index (exact)"])
        vfull, hfull = vp.field_data(in_field_ratio=1.0, slope_exclusion_ratio=ser), hp.field_data(in_field_ratio=1.0, slope_exclusion_ratio=ser)
        res.update(field_size_vertical_mm=vfull["width (exact) mm"], field_size_horizontal_mm=hfull["width (exact) mm"],
                   beam_center_to_top_mm=vfull["left distance->beam center (exact) mm"],
                   beam_center_to_bottom_mm=vfull["right distance->beam center (exact) mm"],
                   beam_center_to_left_mm=hfull["left distance->beam center (exact) mm"],
                   beam_center_to_right_mm=hfull["right distance->beam center (exact) mm"],
                   cax_to_top_mm=vfull["left distance->CAX (exact) mm"], cax_to_bottom_mm=vfull["right distance->CAX (exact) mm"],
                   cax_to_left_mm=hfull["left distance->CAX (exact) mm"], cax_to_right_mm=hfull["right distance->CAX (exact) mm"])
        hfd, vfd = hp.field_data(in_field_ratio=ifr, slope_exclusion_ratio=ser), vp.field_data(in_field_ratio=ifr, slope_exclusion_ratio=ser)
        res.update(top_position_index_x_y=(hfd['"top" index (exact)'], vfd['"top" index (exact)']),
                   top_horizontal_distance_from_cax_mm=hfd['"top"->CAX (exact) mm'], top_vertical_distance_from_cax_mm=vfd['"top"->CAX (exact) mm'],
                   top_horizontal_distance_from_beam_center_mm=hfd['"top"->beam center (exact) mm'],
                   top_vertical_distance_from_beam_center_mm=vfd['"top"->beam center (exact) mm'],
                   left_slope_percent_mm=hfd["left slope (%/mm)"], right_slope_percent_mm=hfd["right slope (%/mm)"],
                   top_slope_percent_mm=vfd["left slope (%/mm)"], bottom_slope_percent_mm=vfd["right slope (%/mm)"])
        want_keys = {k.split(".", 2)[2] for k in g.files if k.startswith(f"{n}.results.")}
        assert want_keys == set(res), (n, want_keys ^ set(res))
        for k, v in res.items():
            assert np.allclose(np.asarray(v, float).reshape(-1), g[f"{n}.results.{k}"], equal_nan=True, **tol), (n, k, v, g[f"{n}.results.{k}"])
        for name, calc in protocols[kw["protocol"]].items():
            for tag, prof in (("horizontal", hp), ("vertical", vp)):
                got = calc(prof, ifr, slope_exclusion_ratio=ser)
                assert np.allclose(got, g[f"{n}.protocol.{name}_{tag}"], **tol), (n, name, tag)
