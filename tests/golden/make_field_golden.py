"""Golden vectors for the strip-profile extraction and centre search of FieldAnalysis (SURVEY.md section 8 row a7:
pylinac/field_analysis.py:488-506, 1068-1117), produced by the reference's OWN methods bound to a stand-in analyzer that
only carries `.image`.  Build container only:

    python tests/golden/make_field_golden.py        # -> tests/golden/field_strips.npz
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref_loader  # noqa: E402
from make_golden import synth_frames  # noqa: E402

warnings.filterwarnings("ignore")
fa = ref_loader.ref("field_analysis")
image = ref_loader.ref("core.image")
prof = ref_loader.ref("core.profile")

frames = synth_frames(2, 300, 400, seed=55)
frames[1] = np.roll(frames[1], (17, -23), (0, 1))
out = {"frames": frames, "frames_f64": frames.astype(np.float64) * 0.731 + 3.25}
specs = np.array([[0.5, 0.03], [0.5, 0.001], [0.3, 0.2], [0.01, 0.1], [0.99, 0.08], [0.62, 0.0]])
out["specs"] = specs
for name in ("frames", "frames_f64"):
    for i, arr in enumerate(out[name]):
        stand_in = types.SimpleNamespace(image=image.ArrayImage(arr.copy(), dpi=100))
        for k, (pos, width) in enumerate(specs):
            hv, b, t = fa.FieldAnalysis._get_horiz_values(stand_in, pos, width)
            vv, l, r = fa.FieldAnalysis._get_vert_values(stand_in, pos, width)
            out[f"{name}.{i}.h{k}"], out[f"{name}.{i}.v{k}"] = np.asarray(hv, float), np.asarray(vv, float)
            out[f"{name}.{i}.edges{k}"] = np.array([b, t, l, r])
        for cname, c in (("beam", prof.Centering.BEAM_CENTER), ("geo", prof.Centering.GEOMETRIC_CENTER)):
            out[f"{name}.{i}.center_{cname}"] = np.array(fa.FieldAnalysis._determine_center(stand_in, c), dtype=float)
np.savez_compressed(os.path.join(HERE, "field_strips.npz"), **out)
print({k: out[k] for k in out if "center" in k})
