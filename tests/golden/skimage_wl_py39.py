"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN per-image Winston-Lutz sequence
(WLBaseImage.analyze, pylinac/winston_lutz.py:709-725): check_inversion_by_histogram((0.01, 50, 99.99)) -> _clean_edges ->
ground -> normalize -> find_field_centroids -> find_bb_centroids (BaseImage.compute + SizedDiskLocator.from_center_physical
-> metrics.utils.find_features), driven on an ArrayImage that borrows those methods unchanged.  Build container only:

    /opt/conda/bin/python3.9 tests/golden/skimage_wl_py39.py tests/golden/wl.npz /root/repo
"""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[2])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
from skimage.measure._regionprops import RegionProperties

RegionProperties.area_filled = property(lambda self: self.filled_area)       # the reference uses the >=0.19 names
RegionProperties.area_bbox = property(lambda self: self.bbox_area)
image = rl.ref("core.image")
wl = rl.ref("winston_lutz")
import importlib.util

_spec = importlib.util.spec_from_file_location("pl_synthetic", sys.argv[2] + "/pylinac_amd/synthetic.py")   # no package init
_syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_syn)
wl_frames = _syn.wl_frames


class W(image.ArrayImage):
    """an array image that borrows WLBaseImage's per-image methods unchanged"""
    detection_conditions = wl.WinstonLutz2D.detection_conditions      # winston_lutz.py:1144-1150


for name in ("_clean_edges", "find_field_centroids", "find_bb_centroids", "_calculate_bb_tolerance"):
    setattr(W, name, getattr(wl.WLBaseImage, name))

PIXEL_MM = 0.336
frames = wl_frames(10, 512, 640, seed0=3000, pixel_mm=PIXEL_MM)              # config #4 recipe, smaller frame
rng = np.random.default_rng(5)
frames = frames.astype(np.int64)
frames[:6] += rng.integers(0, 60, frames[:6].shape)                           # detector noise floor on some frames
frames[6] = 60000 - frames[6]                                                 # inverted polarity -> the inversion branch
frames[7] = frames[7] * 6 // 10                                               # plateau 39 000: a saturated edge is > 10 % above it
frames[7, :2, 100:300] = 65535                                                # dirty edges (< 0.5 % of the frame) ->
frames[7, 50:150, -3:] = 65535                                                # _clean_edges crops twice
frames[8] = np.roll(frames[8], (9, -7), axis=(0, 1))                          # off-centre field
frames = np.clip(frames, 0, 65535).astype(np.uint16)
out = {"frames": frames, "pixel_mm": np.array(PIXEL_MM), "bb_mm": np.array(5.0)}
rec, shapes, inverted = [], [], []
for k, f in enumerate(frames):
    img = W(f.copy(), dpi=25.4 / PIXEL_MM)
    inverted.append(bool(img.check_inversion_by_histogram(percentiles=(0.01, 50, 99.99))))
    img._clean_edges()
    shapes.append(img.array.shape)
    img.ground()
    img.normalize()
    fld = img.find_field_centroids(is_open_field=False)[0]
    try:
        bbs = img.find_bb_centroids(bb_diameter_mm=5.0, low_density=False)
        bb = (bbs[0].x, bbs[0].y, len(bbs))
    except ValueError:
        bb = (np.nan, np.nan, 0)
    rec.append([fld.x, fld.y, bb[0], bb[1], bb[2]])
out["record"] = np.array(rec, dtype=float)
out["shape_after_clean"] = np.array(shapes, dtype=np.int64)
out["inverted"] = np.array(inverted)
np.savez_compressed(sys.argv[1], **out)
print(out["record"], out["shape_after_clean"], out["inverted"])
