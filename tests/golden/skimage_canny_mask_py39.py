"""Helper run under /opt/conda/bin/python3.9: scikit-image 0.18.3's feature.canny WITH a mask (smooth_with_function_and_mask +
the eroded mask, skimage/feature/_canny.py) on float64 and uint16 images.  Build container only.
    /opt/conda/bin/python3.9 tests/golden/skimage_canny_mask_py39.py tests/golden/canny_mask.npz"""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from scipy import ndimage
from skimage import feature

rng = np.random.default_rng(29)
out = {}


def blob(shape, amp):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(float)
    img = float(amp) * (np.hypot(yy - shape[0] * 0.45, xx - shape[1] * 0.55) < min(shape) * 0.3)
    img += 0.5 * amp * ((np.abs(yy - shape[0] * 0.7) < 6) & (np.abs(xx - shape[1] * 0.3) < 14))
    return ndimage.gaussian_filter(img, 1.2) + rng.normal(0, amp * 0.01, shape) + amp * 0.05


def disk(shape, cy, cx, r):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(float)
    return np.hypot(yy - shape[0] * cy, xx - shape[1] * cx) < min(shape) * r


cases = [
    ("f64_disk", blob((70, 90), 1.0), disk((70, 90), 0.5, 0.5, 0.42), dict(sigma=2, low_threshold=0.1, high_threshold=0.6, use_quantiles=True)),
    ("f64_half", blob((64, 64), 1.0), np.mgrid[0:64, 0:64][1] < 40, dict(sigma=1.5, low_threshold=0.02, high_threshold=0.08)),
    ("f64_speckle", blob((50, 60), 1.0), rng.random((50, 60)) < 0.9, dict(sigma=1.0)),
    ("f64_border", blob((48, 52), 1.0), np.pad(np.ones((44, 48), bool), 2), dict(sigma=1.0, low_threshold=0.05, high_threshold=0.1)),
    ("u16_disk", np.clip(blob((72, 66), 30000), 0, 65535).astype(np.uint16), disk((72, 66), 0.45, 0.55, 0.45),
     dict(sigma=1.5, low_threshold=900.0, high_threshold=2500.0)),
    ("f64_allfalse", blob((20, 24), 1.0), np.zeros((20, 24), bool), dict(sigma=1.0)),
]
out["names"] = np.array([c[0] for c in cases])
for name, img, mask, kw in cases:
    out[name + ".img"] = img
    out[name + ".mask"] = mask
    out[name + ".kw"] = np.array(repr(kw))
    out[name + ".edges"] = feature.canny(img, mask=mask, **kw)
    print(name, int(out[name + ".edges"].sum()))
np.savez_compressed(sys.argv[1], **out)
