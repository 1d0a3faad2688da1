"""Golden vectors for the Starshot per-image measurement (the caller of rows a5-a12 that
north_star names), produced by the reference's OWN pylinac.starshot.Starshot.analyze() on synthetic star-shot frames
(the analyzer object is built around an ArrayImage: the reference only loads files).  Build container only:

    python tests/golden/make_starshot_golden.py        # -> tests/golden/starshot.npz
"""
import os
import sys
import warnings

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402

warnings.filterwarnings("ignore")
ss = ref_loader.ref("starshot")
image = ref_loader.ref("core.image")
rw = ref_loader.ref("core.warnings")


def star_frame(shape, centre, n_spokes, width, offsets, seed, amp=3000.0, base=200.0, noise=15.0, inverted=False,
               first_angle=0.17, dtype=np.uint16):
    """n_spokes radiation lines through `centre` (each shifted sideways by offsets[k] px), blurred, noisy"""
    rng = np.random.default_rng(seed)
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w].astype(float)
    img = np.zeros(shape)
    for k in range(n_spokes):
        a = np.pi * k / n_spokes + first_angle
        d = (xx - centre[0]) * np.sin(a) - (yy - centre[1]) * np.cos(a) - offsets[k % len(offsets)]
        img += np.exp(-0.5 * (d / width) ** 2)
    img = ndimage.gaussian_filter(img, 1.0) * amp + base + rng.normal(0, noise, shape)
    if inverted:
        img = img.max() + img.min() - img
    if np.issubdtype(dtype, np.integer):
        return np.clip(img, 0, np.iinfo(dtype).max).astype(dtype)
    return img.astype(dtype)


CASES = {
    "four": dict(frame=dict(shape=(600, 640), centre=(310.4, 295.7), n_spokes=4, width=3.0, offsets=(0.0, 1.5, -1.0, 0.5), seed=0),
                 dpi=100, kw=dict()),
    "six_off": dict(frame=dict(shape=(720, 700), centre=(330.2, 371.9), n_spokes=6, width=2.2, offsets=(0.8, -0.6, 0.3, -1.1, 0.0, 0.9), seed=1),
                    dpi=150, kw=dict(radius=0.7, min_peak_height=0.3)),
    "inverted": dict(frame=dict(shape=(512, 512), centre=(250.5, 262.25), n_spokes=5, width=2.5, offsets=(0.4, -0.4, 1.2, 0.0, -0.9), seed=2, inverted=True),
                     dpi=72, kw=dict()),
    "nofwhm": dict(frame=dict(shape=(640, 600), centre=(301.0, 322.0), n_spokes=4, width=4.0, offsets=(0.0, 0.7, -0.7, 0.2), seed=3, first_angle=0.4),
                   dpi=100, kw=dict(fwhm=False, radius=0.6)),
    "float": dict(frame=dict(shape=(500, 540), centre=(268.7, 251.3), n_spokes=3, width=3.0, offsets=(0.0, 2.0, -1.5), seed=4, dtype=np.float32),
                  dpi=96, kw=dict(tolerance=0.4)),
    "startpt": dict(frame=dict(shape=(600, 640), centre=(310.4, 295.7), n_spokes=4, width=3.0, offsets=(0.0, 1.5, -1.0, 0.5), seed=0),
                    dpi=100, kw=dict(start_point=(318, 290), recursive=False)),
}

out = {"names": np.array(list(CASES))}
for name, c in CASES.items():
    arr = star_frame(**c["frame"])
    s = object.__new__(ss.Starshot)
    rw.WarningCollectorMixin.__init__(s)
    s.image = image.ArrayImage(arr.copy(), dpi=c["dpi"], sid=1000)
    s.wobble = ss.Wobble()
    s.tolerance = 1
    s.analyze(**c["kw"])
    if name != "startpt":          # "startpt" analyses the frame of "four" from a manual start point
        out[f"{name}.frame"] = arr
    out[f"{name}.dpi"] = np.float64(c["dpi"])
    kw = dict(c["kw"])
    out[f"{name}.kw"] = np.array(repr(kw))
    cp = s.circle_profile
    out[f"{name}.profile"] = np.asarray(cp.values, float)
    out[f"{name}.circle"] = np.array([cp.center.x, cp.center.y, cp.radius], dtype=float)
    out[f"{name}.peaks"] = np.array([[p.idx, p.value, p.x, p.y] for p in cp.peaks], dtype=float)
    out[f"{name}.lines"] = np.array([[ln.point1.x, ln.point1.y, ln.point2.x, ln.point2.y] for ln in s.lines.lines], dtype=float)
    out[f"{name}.wobble"] = np.array([s.wobble.center.x, s.wobble.center.y, s.wobble.radius, s.wobble.radius_mm,
                                      s.wobble.diameter_mm], dtype=float)
    out[f"{name}.angles"] = np.array(s.angles, dtype=float)
    out[f"{name}.passed"] = np.array(bool(s.passed))
    sp, local_max = object.__getattribute__(s, "_get_reasonable_start_point")()
    out[f"{name}.start"] = np.array([sp.x, sp.y, local_max], dtype=float)
    print(name, out[f"{name}.wobble"], len(s.lines.lines), out[f"{name}.start"])
np.savez_compressed(os.path.join(HERE, "starshot.npz"), **out)
