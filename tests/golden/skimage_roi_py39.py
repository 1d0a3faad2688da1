"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN DiskROI
(pylinac/core/roi.py:38-140) on a CT-like slice.  Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[3])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
roi = rl.ref("core.roi")
geo = rl.ref("core.geometry")

d = np.load(sys.argv[1])
out = {}
for name in ("slice_i16", "slice_f64"):
    arr = d[name]
    rows = []
    for cx, cy, r in d["rois"]:
        m = roi.DiskROI(arr, radius=float(r), center=geo.Point(float(cx), float(cy)))
        rows.append([len(m.circle_mask()), m.mean, m.std, m.min, m.max, m.pixel_value])
    out["stats_" + name[6:]] = np.array(rows, dtype=float)
pc = roi.DiskROI.from_phantom_center(d["slice_i16"], angle=30.0, roi_radius=7.5, dist_from_center=60.25,
                                     phantom_center=geo.Point(250.3, 260.7))
out["from_center"] = np.array([pc.center.x, pc.center.y, pc.mean, pc.std, pc.pixel_value])
np.savez_compressed(sys.argv[2], **out)
