"""Helper run under /opt/conda/bin/python3.9: the reference's OWN pylinac.ct.ThicknessROI (ct.py:300-313: Gaussian(1) of
the unrotated rectangle window, maximum along the short axis, FWHM of that profile) on synthetic wire-ramp slices.
Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from scipy import ndimage
from skimage import transform

# see skimage_rect_py39.py: scikit-image 0.18.3's matrix_transform wants the matrix, not the transform object
_mt = transform.matrix_transform
transform.matrix_transform = lambda coords, m: _mt(coords, getattr(m, "params", m))

sys.path.insert(0, sys.argv[2])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm", "matplotlib", "PIL", "webbrowser"])
ct = rl.ref("ct")
geo = rl.ref("core.geometry")

rng = np.random.default_rng(41)
h = w = 200
yy, xx = np.mgrid[0:h, 0:w].astype(float)
out = {}
rows = []
specs = []
for k, (cx, cy, width, height, ramp_len, horiz) in enumerate([(100.0, 60.0, 40.0, 10.0, 17.0, True), (60.5, 110.25, 12.0, 44.0, 23.5, False),
                                                               (140.0, 130.0, 36.0, 8.0, 9.0, True), (90.0, 160.0, 10.0, 30.0, 12.0, False)]):
    img = np.full((h, w), 90.0)
    if horiz:      # a bright wire segment along x inside the ROI
        img += 400.0 * ((np.abs(yy - cy) < 1.2) & (np.abs(xx - cx) < ramp_len / 2))
    else:
        img += 400.0 * ((np.abs(xx - cx) < 1.2) & (np.abs(yy - cy) < ramp_len / 2))
    img = ndimage.gaussian_filter(img, 0.8) + rng.normal(0, 2.0, (h, w))
    arr = img.astype(np.int16) if k % 2 == 0 else img
    roi = ct.ThicknessROI(arr, width=width, height=height, center=geo.Point(cx, cy))
    out[f"img{k}"] = arr
    specs.append([cx, cy, width, height])
    rows.append([roi.wire_fwhm, len(roi.long_profile.values), float(np.max(roi.long_profile.values))])
    out[f"profile{k}"] = np.asarray(roi.long_profile.values, dtype=float)
out["specs"] = np.array(specs)
out["results"] = np.array(rows, dtype=float)
np.savez_compressed(sys.argv[1], **out)
print(out["results"])
