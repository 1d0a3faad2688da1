"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): drives the reference's OWN
pylinac.metrics.utils.find_features (pylinac/metrics/utils.py:66-190) and its predicates
(pylinac/metrics/features.py) on BB windows, and records the per-threshold regionprops the sweep saw.
Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[3])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
from skimage import measure, segmentation
from skimage.measure._regionprops import RegionProperties

# the reference uses the >=0.19 attribute names
RegionProperties.area_filled = property(lambda self: self.filled_area)
RegionProperties.area_bbox = property(lambda self: self.bbox_area)
mu = rl.ref("metrics.utils")
ft = rl.ref("metrics.features")
au = rl.ref("core.array_utils")

d = np.load(sys.argv[1])
dpmm = float(d["dpmm"])
radius_mm, tol_mm = float(d["radius_mm"]), float(d["tol_mm"])
conds = [ft.is_right_size_bb, ft.is_round, ft.is_right_circumference, ft.is_symmetric, ft.is_solid]
out = {}
for k in range(int(d["count"])):
    window = d[f"w{k}"]
    sample = au.invert(window)                      # SizedDiskRegion.calculate, metrics/image.py:594-595
    try:
        pts, _, regions = mu.find_features(sample, top_offset=0, left_offset=0, min_number=1,
                                           max_number=int(d["maxn"][k]), dpmm=dpmm,
                                           detection_conditions=conds, radius_mm=radius_mm,
                                           radius_tolerance_mm=tol_mm, min_separation_mm=float(d["minsep"][k]))
        out[f"{k}.points"] = np.array([[p.x, p.y] for p in pts], dtype=float)
    except ValueError:
        out[f"{k}.points"] = np.zeros((0, 2))
    # per-level tables of what the sweep sees (for pinning the restated regionprops)
    s = au.stretch(sample, min=0, max=1)
    rows = []
    cutoff = 0.0 + 1.0 / 50
    for lvl in range(50):
        if cutoff > 1.0:
            break
        lab = segmentation.clear_border(measure.label(s > cutoff, connectivity=1))
        for r in measure.regionprops(lab, intensity_image=s):
            if r.area < 4:
                continue
            rows.append([lvl, r.label, r.area, r.filled_area, *r.bbox, r.perimeter, r.convex_area, r.solidity,
                         *r.weighted_centroid])
        cutoff += 1.0 / 50
    out[f"{k}.levels"] = np.array(rows, dtype=float).reshape(-1, 13)
np.savez_compressed(sys.argv[2], **out)
