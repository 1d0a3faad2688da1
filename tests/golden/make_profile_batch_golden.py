"""Golden vectors for the BATCHED SingleProfile paths (profile.single_profile_inflection_batch / _fwhm_batch / _hill_batch:
equal-length profiles with index abscissae), produced by the reference's OWN pylinac.core.profile.SingleProfile on its frozen
63-detector regression profiles and on synthetic open-field profiles.  Build container only:

    python tests/golden/make_profile_batch_golden.py        # -> tests/golden/profile_batch.npz
"""
import importlib.util
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_loader  # noqa: E402
from next_row_checks import beam_profiles  # noqa: E402

warnings.filterwarnings("ignore")
prof = ref_loader.ref("core.profile")
spec = importlib.util.spec_from_file_location(
    "profile_regression_fixtures", "/root/reference/tests_basic/core/profile_regression_fixtures.py")
fxm = importlib.util.module_from_spec(spec)
sys.modules["profile_regression_fixtures"] = fxm
spec.loader.exec_module(fxm)

INFL_KEYS = ["left index (exact)", "right index (exact)", "left value (@rounded)", "left value (@exact)",
             "right value (@rounded)", "right value (@exact)"]
FWHM_PEN = ["left 20% index (exact)", "left 80% index (exact)", "right 20% index (exact)", "right 80% index (exact)",
            "left 20% value (@rounded)", "right 80% value (@rounded)", "left penumbra width (exact)", "right penumbra width (exact)"]
HILL_PEN = ["left 20% index (exact)", "left 80% index (exact)", "right 20% index (exact)", "right 80% index (exact)",
            "left 20% value (exact)", "right 80% value (exact)", "left penumbra width (exact)", "right penumbra width (exact)",
            "left gradient (exact)", "right gradient (exact)"]
out = {"infl_keys": np.array(INFL_KEYS), "fwhm_pen_keys": np.array(FWHM_PEN), "hill_pen_keys": np.array(HILL_PEN)}

fixtures = [f for f in fxm.PROFILE_REGRESSION_FIXTURES if len(f.values) == 63]
sets = {"fx63": np.stack([np.asarray(f.values, float) for f in fixtures]),
        "beam": beam_profiles(6, 120, seed=23)}
OPTS = {"default": {}, "none": dict(interpolation=prof.Interpolation.NONE), "max_dpmm": dict(normalization_method=prof.Normalization.MAX, dpmm=2.0)}
for sname, rows in sets.items():
    if rows is None:
        continue
    out[f"{sname}.rows"] = rows
    for oname, kw in OPTS.items():
        for r, row in enumerate(rows):
            tag = f"{sname}.{oname}.{r}"
            for edge, enum in (("infl", prof.Edge.INFLECTION_DERIVATIVE), ("fwhm", prof.Edge.FWHM), ("hill", prof.Edge.INFLECTION_HILL)):
                try:
                    p = prof.SingleProfile(row.copy(), edge_detection_method=enum, **kw)
                    if edge == "infl":
                        d = p.inflection_data()
                        out[f"{tag}.infl"] = np.array([float(d[k]) for k in INFL_KEYS])
                        out[f"{tag}.infl_values"] = np.asarray(p.values, float)
                    elif edge == "fwhm":
                        d = p.penumbra(20, 80)
                        out[f"{tag}.fwhm_pen"] = np.array([float(d[k]) for k in FWHM_PEN])
                    else:
                        d = p.penumbra(20, 80)
                        out[f"{tag}.hill_pen"] = np.array([float(d[k]) for k in HILL_PEN])
                except Exception as exc:      # the reference raises for this profile: recorded, the batch must report it
                    out[f"{tag}.{edge}.error"] = np.array(type(exc).__name__)
out["sets"] = np.array([k for k, v in sets.items() if v is not None])
out["opts"] = np.array(list(OPTS))
np.savez_compressed(os.path.join(HERE, "profile_batch.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:6]}, len(out))
print([k for k in out if k.endswith("error")][:10])
