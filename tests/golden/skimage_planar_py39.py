"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the planar phantom outline search of
pylinac/planar_imaging.py:300-341, 574-588, 3136-3179 done with scikit-image's own canny / label / regionprops /
hough_line / hough_line_peaks on synthetic phantom frames.  Build container only."""
import sys, warnings
import numpy as np
warnings.filterwarnings("ignore")
from skimage import feature, measure, transform
from scipy import ndimage

rng = np.random.default_rng(11)
out = {}


def phantom(shape, half, angle_deg, centre=None, contrast=0.5, noise=0.01, blur=1.5):
    """a rotated square slab on a flat background, blurred and noisy"""
    h, w = shape
    cy, cx = centre if centre else ((h - 1) / 2, (w - 1) / 2)
    yy, xx = np.mgrid[0:h, 0:w].astype(float)
    a = np.deg2rad(angle_deg)
    u = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a)
    v = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a)
    img = 0.2 + contrast * ((np.abs(u) < half) & (np.abs(v) < half))
    img = ndimage.gaussian_filter(img, blur)
    return img + rng.normal(0, noise, shape)


cases = [
    ("sq45", phantom((160, 176), 38, 45.0), dict(sigma=2, low=0.001, high=0.01)),
    ("sq43", phantom((150, 150), 34, 43.2, centre=(70.3, 78.1)), dict(sigma=2, low=0.001, high=0.01)),
    ("sq0", phantom((128, 144), 36, 0.0, noise=0.004), dict(sigma=4, low=0.001, high=0.01)),
]
out["names"] = np.array([c[0] for c in cases])
for name, img, kw in cases:
    edges = feature.canny(img, low_threshold=kw["low"], high_threshold=kw["high"], use_quantiles=True, sigma=kw["sigma"])
    lab = measure.label(edges)
    regions = measure.regionprops(lab, intensity_image=img)
    bbox = np.array([r.bbox for r in regions], dtype=np.int64).reshape(-1, 4)
    out[name + ".img"] = img
    out[name + ".kw"] = np.array([kw["sigma"], kw["low"], kw["high"]], dtype=float)
    out[name + ".edges"] = edges
    out[name + ".bbox"] = bbox
    out[name + ".bbox_area"] = np.array([r.bbox_area for r in regions], dtype=np.int64)
    # the region the reference would keep for a phantom of this size: biggest bbox (used for the Hough stage below)
    big = int(np.argmax([r.bbox_area for r in regions]))
    out[name + ".big"] = np.array(big)
    r = regions[big]
    out[name + ".region_image"] = r.image
    out[name + ".orientation"] = np.array(r.orientation)
    out[name + ".centroid"] = np.array(r.centroid)
    out[name + ".intensity_minmax"] = np.array([r.intensity_image[r.image].min(), r.intensity_image[r.image].max(),
                                               np.min(r.intensity_image), np.max(r.intensity_image)])
    # planar_imaging.py:3136-3166 (angle band 40..50 deg at 0.01 deg is 1001 columns: use 0.05 deg here to keep the file small)
    theta = np.deg2rad(np.linspace(40, 50, 201))
    hs, an, di = transform.hough_line(r.image, theta=theta)
    for md in (int(70 * 0.25), 9):
        for npk in (2, np.inf):
            ph, pa, pd = transform.hough_line_peaks(hs, an, di, min_distance=md, num_peaks=npk)
            tag = f"{name}.peaks.md{md}.n{'inf' if npk == np.inf else npk}"
            out[tag + ".h"] = np.asarray(ph)
            out[tag + ".a"] = np.asarray(pa, dtype=float)
            out[tag + ".d"] = np.asarray(pd, dtype=float)
    out[name + ".hspace"] = hs
    out[name + ".theta"] = theta
    out[name + ".dists"] = di

# hough_line_peaks on accumulators with plateaus, ties, wrap-around columns and the default full sweep
for k in range(4):
    shape = [(40, 36), (64, 90), (30, 181), (25, 12)][k]
    hs = rng.integers(0, 40, shape).astype(np.uint64)
    if k == 1:
        hs[10:13, 20:22] = 77          # a plateau -> one multi-pixel candidate group
        hs[40, 0] = 90                 # column 0: suppression wraps to the far columns
        hs[50, 89] = 90
    if k == 2:
        hs = (hs // 8) * 8             # many ties
    an = np.linspace(-np.pi / 2, np.pi / 2, shape[1], endpoint=False)
    di = np.linspace(-shape[0] / 2, shape[0] / 2, shape[0])
    out[f"acc{k}.hspace"] = hs
    out[f"acc{k}.angles"] = an
    out[f"acc{k}.dists"] = di
    for j, kw in enumerate([dict(), dict(min_distance=3, min_angle=4), dict(min_distance=1, min_angle=30, num_peaks=3),
                            dict(min_distance=5, min_angle=2, threshold=20.0)]):
        ph, pa, pd = transform.hough_line_peaks(hs, an, di, **kw)
        out[f"acc{k}.kw{j}"] = np.array([kw.get("min_distance", 9), kw.get("min_angle", 10), kw.get("threshold", -1.0),
                                        kw.get("num_peaks", -1)], dtype=float)
        out[f"acc{k}.p{j}.h"] = np.asarray(ph)
        out[f"acc{k}.p{j}.a"] = np.asarray(pa, dtype=float)
        out[f"acc{k}.p{j}.d"] = np.asarray(pd, dtype=float)
np.savez_compressed(sys.argv[1], **out)
