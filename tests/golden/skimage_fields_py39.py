"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): drives the reference's OWN
GlobalSizedFieldLocator.calculate (pylinac/metrics/image.py:817-897) on synthetic multi-field frames.
Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[3])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
from skimage.measure._regionprops import RegionProperties

# the reference uses the >=0.19 attribute names
RegionProperties.area_filled = property(lambda self: self.filled_area)
RegionProperties.area_bbox = property(lambda self: self.bbox_area)
RegionProperties.equivalent_diameter_area = property(lambda self: self.equivalent_diameter)
mi = rl.ref("metrics.image")


class FakeImage:
    def __init__(self, array, dpmm):
        self.array, self.dpmm = array, dpmm


d = np.load(sys.argv[1])
out = {}
for k in range(int(d["count"])):
    loc = mi.GlobalSizedFieldLocator.from_physical(field_width_mm=float(d["fw"][k]), field_height_mm=float(d["fh"][k]),
                                                   field_tolerance_mm=float(d["tol"][k]), max_number=int(d["maxn"][k]))
    loc.image = FakeImage(d[f"f{k}"], float(d["dpmm"]))
    try:
        pts = loc.calculate()
        out[f"{k}.points"] = np.array([[p.x, p.y] for p in pts], dtype=float).reshape(-1, 2)
    except ValueError:
        out[f"{k}.points"] = np.zeros((0, 2))
np.savez_compressed(sys.argv[2], **out)
