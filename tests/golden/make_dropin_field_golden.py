"""Golden for the GPU replay of FieldAnalysis (tests/test_gpu_dropin.py): the reference's OWN ``FieldAnalysis.analyze()``
(pylinac/field_analysis.py:562-965) on a synthetic open field, several protocol / centering / edge-detection / interpolation
combinations -- every entry of ``_results`` and ``_extra_results`` and both processed profiles.  Build container only:

    python tests/golden/make_dropin_field_golden.py        # -> tests/golden/dropin_field.npz
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402

warnings.filterwarnings("ignore")
fa = ref_loader.ref("field_analysis")
image = ref_loader.ref("core.image")

CASES = [dict(protocol="VARIAN"),
         dict(protocol="ELEKTA", edge_detection_method="FWHM", centering="Geometric center", vert_width=0.05, horiz_width=0.05),
         dict(protocol="SIEMENS", edge_detection_method="Inflection Hill", is_FFF=True, interpolation="Spline",
              normalization_method="Max", hill_window_ratio=0.1),
         dict(protocol="NONE", centering="Manual", vert_position=0.45, horiz_position=0.55, interpolation=None)]

g = np.load(os.path.join(HERE, "field_strips.npz"))
arr = g["frames"][1]
out = {"frame": arr, "dpi": np.float64(100), "cases": np.array(json.dumps(CASES))}
for n, kw in enumerate(CASES):
    kw = dict(kw)
    kw["protocol"] = getattr(fa.Protocol, kw["protocol"])
    f = object.__new__(fa.FieldAnalysis)                 # FieldAnalysis.__init__ (:448-470) minus image.load
    f._path, f._is_analyzed, f._from_device = "array", False, False
    f.image = image.ArrayImage(arr.copy(), dpi=100)
    f.image.check_inversion_by_histogram()
    f.analyze(**kw)
    for k, v in f._results.items():
        out[f"{n}.results.{k}"] = np.asarray(v, dtype=float).reshape(-1)
    for k, v in f._extra_results.items():
        out[f"{n}.protocol.{k}"] = np.asarray(v, dtype=float).reshape(-1)
    out[f"{n}.horiz"], out[f"{n}.vert"] = np.asarray(f.horiz_profile.values, float), np.asarray(f.vert_profile.values, float)
    print(n, kw["protocol"].name, len(f._results), "results,", len(f._extra_results), "protocol values")
np.savez_compressed(os.path.join(HERE, "dropin_field.npz"), **out)
