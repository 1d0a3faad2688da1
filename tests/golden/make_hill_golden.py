"""Golden vectors for SingleProfile's Hill-fit edge method and penumbra() (SURVEY.md section 8 "next" row f4, second
half), produced by the reference's OWN pylinac.core.profile.SingleProfile / pylinac.core.hill.Hill (real scipy
curve_fit underneath) on its 20 frozen regression profiles and an EPID-style profile.  Build container only:

    python tests/golden/make_hill_golden.py        # -> tests/golden/hill.npz
"""
import importlib.util
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref_loader  # noqa: E402
from make_golden import synth_frames  # noqa: E402

warnings.filterwarnings("ignore")
prof = ref_loader.ref("core.profile")
spec = importlib.util.spec_from_file_location(
    "profile_regression_fixtures", "/root/reference/tests_basic/core/profile_regression_fixtures.py")
fxm = importlib.util.module_from_spec(spec)
sys.modules["profile_regression_fixtures"] = fxm
spec.loader.exec_module(fxm)

HILL_KEYS = ["left index (exact)", "right index (exact)", "left value (@exact)", "right value (@exact)"]
FIELD_KEYS = ["width (exact)", "beam center index (exact)", "beam center value (@rounded)", "cax index (exact)",
              "left index (exact)", "left slope", "right slope", "right index (exact)"]
out = {"hill_keys": np.array(HILL_KEYS), "field_keys": np.array(FIELD_KEYS)}


def pen_keys(edge, lower, upper, dpmm):
    k = [f"left {lower}% index (exact)", f"left {upper}% index (exact)", f"right {lower}% index (exact)",
         f"right {upper}% index (exact)", "left penumbra width (exact)", "right penumbra width (exact)"]
    if edge == "hill":
        k += [f"left {lower}% value (exact)", f"left {upper}% value (exact)", f"right {lower}% value (exact)",
              f"right {upper}% value (exact)", "left gradient (exact)", "right gradient (exact)"]
    if edge == "fwhm":
        k += [f"left {lower}% value (@rounded)", f"right {upper}% value (@rounded)"]
    if dpmm:
        k += ["left penumbra width (exact) mm", "right penumbra width (exact) mm"]
        if edge == "hill":
            k += ["left gradient (exact) %/mm", "right gradient (exact) %/mm"]
    return k


EDGES = {"fwhm": prof.Edge.FWHM, "infl": prof.Edge.INFLECTION_DERIVATIVE, "hill": prof.Edge.INFLECTION_HILL}


def record(tag, values, edge, **kw):
    try:
        p = prof.SingleProfile(values, edge_detection_method=EDGES[edge], **kw)
        if edge == "hill":
            inf = p.inflection_data()
            out[f"{tag}.infl"] = np.array([float(inf[k]) for k in HILL_KEYS])
            out[f"{tag}.infl_rounded"] = np.array([inf["left index (rounded)"], inf["right index (rounded)"]])
            out[f"{tag}.params"] = np.array([inf["left Hill params"], inf["right Hill params"]], dtype=float)
            bc = p.beam_center()
            out[f"{tag}.beam_center"] = np.array([bc["index (exact)"], bc["value (@rounded)"]], dtype=float)
            fd = p.field_data(in_field_ratio=0.8, slope_exclusion_ratio=0.2)
            out[f"{tag}.field"] = np.array([float(fd[k]) for k in FIELD_KEYS])
        out[f"{tag}.values"] = np.asarray(p.values, float)
        for lower, upper in ((20, 80), (10, 90)):
            pen = p.penumbra(lower, upper)
            keys = pen_keys(edge, lower, upper, kw.get("dpmm"))
            out[f"{tag}.pen{lower}_{upper}"] = np.array([float(pen[k]) for k in keys])
            out[f"{tag}.pen{lower}_{upper}.left_values"] = np.asarray(pen["left values"], float)
            out[f"{tag}.pen{lower}_{upper}.right_values"] = np.asarray(pen["right values"], float)
    except Exception as exc:      # coarse profiles without a usable gradient peak / a failed fit: recorded as such
        out[f"{tag}.error"] = np.array(type(exc).__name__)


out["n_fixtures"] = np.int64(len(fxm.PROFILE_REGRESSION_FIXTURES))
for i, fx in enumerate(fxm.PROFILE_REGRESSION_FIXTURES):
    out[f"fx{i}.x"], out[f"fx{i}.y"] = np.asarray(fx.x_values, float), np.asarray(fx.values, float)
    for edge in EDGES:
        for mode, interp in (("none", prof.Interpolation.NONE), ("linear", prof.Interpolation.LINEAR)):
            if edge == "hill":
                # the Hill branch keeps only window positions >= 0 (profile.py:1683-1691): the fixtures' physical
                # abscissae start below zero, so it is exercised on index positions, with a window wide enough for a
                # four-parameter fit on ~60-sample profiles
                record(f"fx{i}.{edge}.{mode}", fx.values, edge, interpolation=interp, hill_window_ratio=0.5)
            else:
                record(f"fx{i}.{edge}.{mode}", fx.values, edge, x_values=fx.x_values, interpolation=interp)
epid = np.mean(synth_frames(1, 96, 400, seed=91)[0][40:56].astype(float), axis=0)
out["epid.y"] = epid
for edge in EDGES:
    record(f"epid.{edge}.dpmm", epid.copy(), edge, dpmm=1 / 0.336)
    record(f"epid.{edge}.wide", epid.copy(), edge, dpmm=1 / 0.336, hill_window_ratio=0.2, interpolation="Spline")
# flattening-filter-free style profiles (peaked top, soft penumbra), different lengths and sampling
rng = np.random.default_rng(5)
for k, (n, half, soft, peak) in enumerate([(300, 90, 6.0, 0.0), (520, 170, 9.0, 0.4), (255, 60, 3.5, 0.8), (1024, 300, 14.0, 0.6)]):
    x = np.arange(n) - (n - 1) / 2 + rng.uniform(-2, 2)
    y = 1 / (1 + np.exp((np.abs(x) - half) / soft)) * (1 - peak * (np.abs(x) / n) ** 1.3) + rng.normal(0, 0.002, n)
    out[f"fff{k}.y"] = y
    for edge in EDGES:
        record(f"fff{k}.{edge}.px", y.copy(), edge, dpmm=2.5 + k, hill_window_ratio=(0.1, 0.2, 0.3, 0.05)[k])
np.savez_compressed(os.path.join(HERE, "hill.npz"), **out)
errs = {k: str(out[k]) for k in out if k.endswith(".error")}
print(len(out), "arrays;", len(errs), "recorded errors:", sorted(set(errs.values())))
