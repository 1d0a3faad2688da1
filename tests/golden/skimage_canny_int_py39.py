"""Helper run under /opt/conda/bin/python3.9: scikit-image 0.18.3's feature.canny on INTEGER images (uint8 / uint16 /
int16: img_as_float scaling inside skimage.filters.gaussian; absolute thresholds divided by dtype_max).
Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from scipy import ndimage
from skimage import feature

rng = np.random.default_rng(23)
out = {}


def blob(shape, amp):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(float)
    img = float(amp) * (np.hypot(yy - shape[0] * 0.45, xx - shape[1] * 0.55) < min(shape) * 0.3)
    img += 0.5 * amp * ((np.abs(yy - shape[0] * 0.7) < 6) & (np.abs(xx - shape[1] * 0.3) < 14))
    return ndimage.gaussian_filter(img, 1.2) + rng.normal(0, amp * 0.01, shape) + amp * 0.05


cases = [
    ("u16_q", np.clip(blob((70, 90), 40000), 0, 65535).astype(np.uint16), dict(sigma=2, low_threshold=0.001, high_threshold=0.01, use_quantiles=True)),
    ("u16_abs", np.clip(blob((64, 64), 30000), 0, 65535).astype(np.uint16), dict(sigma=1.5, low_threshold=900.0, high_threshold=2500.0)),
    ("u16_default", np.clip(blob((50, 60), 50000), 0, 65535).astype(np.uint16), dict(sigma=1.0)),
    ("u8_q", np.clip(blob((60, 80), 180), 0, 255).astype(np.uint8), dict(sigma=2, low_threshold=0.3, high_threshold=0.8, use_quantiles=True)),
    ("u8_abs", np.clip(blob((60, 80), 180), 0, 255).astype(np.uint8), dict(sigma=1.0, low_threshold=5.0, high_threshold=15.0)),
    ("i16_q", (blob((72, 66), 20000) - 9000).astype(np.int16), dict(sigma=2, low_threshold=0.5, high_threshold=0.9, use_quantiles=True)),
    ("i16_abs", (blob((72, 66), 20000) - 9000).astype(np.int16), dict(sigma=1.0, low_threshold=400.0, high_threshold=1500.0)),
]
out["names"] = np.array([c[0] for c in cases])
for name, img, kw in cases:
    out[name + ".img"] = img
    out[name + ".kw"] = np.array(repr(kw))
    out[name + ".edges"] = feature.canny(img, **kw)
    print(name, int(out[name + ".edges"].sum()))
np.savez_compressed(sys.argv[1], **out)
