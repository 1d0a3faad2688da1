"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN LowContrastDiskROI /
HighContrastDiskROI (pylinac/core/roi.py:186-478) and pylinac.core.contrast functions on the CT-like slices of roi.npz.
Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[3])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
roi = rl.ref("core.roi")
geo = rl.ref("core.geometry")
con = rl.ref("core.contrast")

d = np.load(sys.argv[1])
arr = d["slice_i16"].astype(np.float64) + 1100.0     # positive values, like an EPID / planar image
out = {"shift": np.float64(1100.0)}
PROPS = ["pixel_value", "std", "signal_to_noise", "contrast", "contrast_to_noise", "michelson", "weber", "visibility",
         "cnr_constant", "contrast_constant", "passed", "passed_visibility", "passed_contrast_constant", "passed_cnr_constant"]
out["props"] = np.array(PROPS)
METHODS = ["Michelson", "Weber", "Ratio", "Difference"]
out["methods"] = np.array(METHODS)
cases = [(256.0, 256.0, 10.0, 1150.0), (180.4, 300.7, 7.5, 1200.0), (330.5, 199.5, 12.25, 1188.5), (256.2, 90.9, 30.0, 1250.0)]
out["cases"] = np.array(cases)
rows = []
for cx, cy, r, ref in cases:
    for m in METHODS:
        z = roi.LowContrastDiskROI(arr, radius=r, center=geo.Point(cx, cy), contrast_threshold=0.01,
                                   contrast_reference=ref, cnr_threshold=0.5, contrast_method=m, visibility_threshold=0.1)
        rows.append([float(getattr(z, p)) for p in PROPS])
out["low"] = np.array(rows, dtype=float).reshape(len(cases), len(METHODS), len(PROPS))
fc = roi.LowContrastDiskROI.from_phantom_center(arr, angle=-33.0, roi_radius=6.5, dist_from_center=70.25,
                                                phantom_center=geo.Point(250.3, 260.7), contrast_threshold=0.02,
                                                contrast_reference=1190.0, cnr_threshold=1.0)
out["low_from_center"] = np.array([fc.center.x, fc.center.y, fc.pixel_value, fc.contrast, fc.visibility, float(fc.passed)])
hc = roi.HighContrastDiskROI(arr, radius=9.0, center=geo.Point(200.5, 310.25), contrast_threshold=0.5)
out["high"] = np.array([hc.max, hc.min, hc.pixel_value, hc.std])
# the plain functions
v = np.array([0.2, 0.9, 0.5, 0.33])
out["fn_in"] = v
out["fn"] = np.array([con.michelson(v), con.rms(v), con.weber(3.0, 2.0), con.ratio(3.0, 2.0), con.difference(3.0, 5.5),
                      con.contrast(np.array([3.0, 2.0]), "Weber"), con.contrast(v, "Root Mean Square"),
                      con.visibility(np.array([3.0, 2.0]), 5.0, 0.7, "Michelson")])
np.savez_compressed(sys.argv[2], **out)
