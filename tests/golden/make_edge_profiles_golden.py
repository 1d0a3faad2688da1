"""Golden vectors for the new-style edge-detection profile classes of SURVEY.md section 8 row f4
(pylinac/core/profile.py:612-740: InflectionDerivativeProfile.field_edge_idx :656-670, HillProfile.field_edge_idx
:708-728), produced by the reference's OWN classes on its 20 frozen regression profiles (physical abscissae) and on
EPID / FFF-style profiles.  Build container only:

    python tests/golden/make_edge_profiles_golden.py        # -> tests/golden/edge_profiles.npz
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

out = {}
cases = []


def record(tag, cls, values, **kw):
    try:
        p = cls(np.asarray(values, dtype=float), **kw)
        left, right = p.field_edge_idx("left"), p.field_edge_idx("right")
        out[tag] = np.array([left, right, p.center_idx, p.field_width_px, p.geometric_center_idx, p.cax_index], dtype=float)
    except Exception as exc:
        out[tag + ".error"] = np.array(type(exc).__name__)


for i, fx in enumerate(fxm.PROFILE_REGRESSION_FIXTURES):
    out[f"fx{i}.x"], out[f"fx{i}.y"] = np.asarray(fx.x_values, float), np.asarray(fx.values, float)
    record(f"fx{i}.infl", prof.InflectionDerivativeProfile, fx.values, x_values=np.asarray(fx.x_values, float))
    record(f"fx{i}.infl_ground_max", prof.InflectionDerivativeProfile, fx.values, x_values=np.asarray(fx.x_values, float),
           ground=True, normalization=prof.Normalization.MAX, edge_smoothing_ratio=0.01)
    record(f"fx{i}.hill", prof.HillProfile, fx.values, x_values=np.asarray(fx.x_values, float), hill_window_ratio=0.3)
epid = np.mean(synth_frames(1, 96, 400, seed=91)[0][40:56].astype(float), axis=0)
out["epid.y"] = epid
record("epid.infl", prof.InflectionDerivativeProfile, epid)
record("epid.hill", prof.HillProfile, epid)
record("epid.hill_beam", prof.HillProfile, epid, normalization=prof.Normalization.BEAM_CENTER, hill_window_ratio=0.2)
rng = np.random.default_rng(5)
for k, (n, half, soft, peak) in enumerate([(300, 90, 6.0, 0.0), (520, 170, 9.0, 0.4), (255, 60, 3.5, 0.8)]):
    x = np.arange(n) - (n - 1) / 2 + rng.uniform(-2, 2)
    y = 1 / (1 + np.exp((np.abs(x) - half) / soft)) * (1 - peak * (np.abs(x) / n) ** 1.3) + rng.normal(0, 0.002, n)
    out[f"fff{k}.y"], out[f"fff{k}.x"] = y, x * 0.4
    record(f"fff{k}.infl", prof.InflectionDerivativeProfile, y, x_values=x * 0.4)
    record(f"fff{k}.hill", prof.HillProfile, y, x_values=x * 0.4, hill_window_ratio=0.15)
# ProfileBase.field_x_values / field_values / field_indices / resample_to (profile.py:299-352, 392-431) through FWXMProfile
for i in (0, 3, 7, 12, 19):
    fx = fxm.PROFILE_REGRESSION_FIXTURES[i]
    p = prof.FWXMProfile(np.asarray(fx.values, float), x_values=np.asarray(fx.x_values, float), fwxm_height=50)
    for r in (1.0, 0.8, 0.5):
        out[f"fx{i}.fwxm.field_x.{r}"] = np.asarray(p.field_x_values(r), float)
        out[f"fx{i}.fwxm.field_v.{r}"] = np.asarray(p.field_values(r), float)
        out[f"fx{i}.fwxm.field_idx.{r}"] = np.asarray(p.field_indices(r), float)
    ys = np.array([0.2, 0.5, 0.8]) * (np.max(fx.values) - np.min(fx.values)) + np.min(fx.values)
    out[f"fx{i}.fwxm.x_at_y_in"] = ys
    out[f"fx{i}.fwxm.x_at_y_left"] = np.asarray(p.x_at_y(ys, "left"), float)
    out[f"fx{i}.fwxm.x_at_y_right"] = np.asarray(p.x_at_y(ys, "right"), float)
    tx = np.linspace(np.min(fx.x_values) * 0.6, np.max(fx.x_values) * 0.7, 37)
    q = p.resample_to(prof.FWXMProfile(np.ones(37), x_values=tx))
    out[f"fx{i}.fwxm.resample_x"], out[f"fx{i}.fwxm.resample_y"] = np.asarray(q.x_values, float), np.asarray(q.values, float)
np.savez_compressed(os.path.join(HERE, "edge_profiles.npz"), **out)
errs = [k for k in out if k.endswith(".error")]
print(len(out), "arrays;", len(errs), "errors", sorted({str(out[k]) for k in errs}), errs[:6])
print(out["fx3.infl"], out["fx3.hill"] if "fx3.hill" in out else None, out["epid.hill"])
