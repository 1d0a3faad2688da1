"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): regionprops centroid / orientation / eccentricity /
inertia_tensor of small binary regions, including regions that are symmetric under a swap of the axes (where
``a - c == 0`` decides the +-pi/4 branch of ``orientation``).  Called at pylinac/planar_imaging.py:2348, 2498 and
pylinac/ct.py:2548.  Build container only:  /opt/conda/bin/python3.9 tests/golden/skimage_regionprops_py39.py tests/golden/regionprops.npz"""
import sys
import numpy as np
from skimage import measure

rng = np.random.default_rng(5)
shapes = {}
yy, xx = np.mgrid[0:41, 0:41]
shapes["diamond"] = (np.abs(yy - 20) + np.abs(xx - 20)) <= 14
shapes["diamond_ring"] = ((np.abs(yy - 20) + np.abs(xx - 20)) <= 14) & ((np.abs(yy - 20) + np.abs(xx - 20)) >= 12)
shapes["square"] = (np.abs(yy - 20) <= 9) & (np.abs(xx - 20) <= 9)
shapes["disk"] = ((yy - 20) ** 2 + (xx - 20) ** 2) <= 100
shapes["diag_bar"] = np.abs(yy - xx) <= 2                       # symmetric under transposition, mu11 > 0
shapes["anti_bar"] = np.abs(yy + xx - 40) <= 2                  # symmetric under anti-transposition, mu11 < 0
shapes["hbar"] = (np.abs(yy - 20) <= 2) & (np.abs(xx - 20) <= 15)
shapes["vbar"] = (np.abs(yy - 20) <= 15) & (np.abs(xx - 20) <= 2)
shapes["ell"] = ((yy >= 5) & (yy <= 30) & (xx >= 5) & (xx <= 9)) | ((yy >= 26) & (yy <= 30) & (xx >= 5) & (xx <= 33))
shapes["single"] = (yy == 7) & (xx == 9)
shapes["pair_h"] = (yy == 7) & ((xx == 9) | (xx == 10))
t = rng.random((41, 41)) < 0.5
blob = measure.label(t, connectivity=2)
shapes["blob"] = blob == np.bincount(blob.ravel())[1:].argmax() + 1
tt = shapes["blob"] | shapes["blob"].T                           # transposition-symmetric irregular region
shapes["blob_sym"] = measure.label(tt, connectivity=2) == 1
for k in range(6):
    a = np.deg2rad(rng.uniform(0, 180))
    cy, cx = rng.uniform(15, 25, 2)
    u = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a)
    v = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a)
    shapes[f"ellipse{k}"] = (u / rng.uniform(8, 14)) ** 2 + (v / rng.uniform(3, 7)) ** 2 <= 1
out = {"names": np.array(list(shapes))}
for name, m in shapes.items():
    lab = np.zeros((64, 72), dtype=np.int32)
    lab[11:52, 17:58] = m.astype(np.int32)
    r = measure.regionprops(lab)[0]
    out[name + ".labels"] = lab
    out[name + ".centroid"] = np.array(r.centroid)
    out[name + ".orientation"] = np.array(r.orientation)
    out[name + ".eccentricity"] = np.array(r.eccentricity)
    out[name + ".inertia_tensor"] = np.array(r.inertia_tensor)
    # degenerate = both a - c and b vanish analytically: the orientation is undefined and scikit-image's value is noise
    out[name + ".area"] = np.array(r.area)
np.savez_compressed(sys.argv[1], **out)
