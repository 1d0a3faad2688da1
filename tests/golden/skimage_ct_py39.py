"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3, scipy 1.7.1): the body of
pylinac.ct.get_regions (pylinac/ct.py:3315-3348, Slice branch) + the region choice of
Slice.phantom_roi (ct.py:414-420) on every slice of an .npz.  Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from scipy import ndimage
from skimage import draw, filters, measure, segmentation

d = np.load(sys.argv[1])
slices = d["slices"]
mm_per_pixel = float(d["mm_per_pixel"])
catphan_size = float(d["catphan_size"])
out = {}
stage = str(d["stage"])
for i, arr in enumerate(slices):
    if stage == "scharr":
        out[f"{i}.scharr"] = filters.scharr(arr.astype(float))
        continue
    # skimage.filters.gaussian(edges, sigma=1) == ndimage.gaussian_filter(mode="nearest") is evaluated
    # by the caller under scipy 1.15.3 (the reference pins scipy>=1.11; this interpreter has 1.7.1,
    # whose correlate1d differs in the last bits)
    edges = d[f"gauss{i}"]
    out[f"{i}.gauss"] = edges
    center_y, center_x = arr.shape[0] / 2 - 0.5, arr.shape[1] / 2 - 0.5       # BaseImage.center
    rr, cc = draw.disk(center=(center_y, center_x), radius=110 / mm_per_pixel, shape=edges.shape)
    m = np.zeros(edges.shape, np.uint8)
    m[rr, cc] = 1
    out[f"{i}.disk"] = m
    otsu = filters.threshold_otsu(edges[rr, cc])
    out[f"{i}.otsu"] = np.float64(otsu)
    thres = otsu * 0.8
    bw = edges > thres
    out[f"{i}.bw"] = bw.astype(np.uint8)
    bw = segmentation.clear_border(bw, buffer_size=min(int(max(bw.shape) / 100), 3))
    out[f"{i}.cleared"] = bw.astype(np.uint8)
    bw = ndimage.binary_fill_holes(bw)
    out[f"{i}.filled"] = bw.astype(np.uint8)
    lab, num = measure.label(bw, return_num=True)
    out[f"{i}.labels"] = lab.astype(np.int32)
    props = measure.regionprops(lab, edges)
    tab = np.array([[p.area, *p.bbox, *p.centroid, p.filled_area, *p.weighted_centroid] for p in props], dtype=float)
    out[f"{i}.props"] = tab.reshape(-1, 10)
    if props:
        best = sorted(props, key=lambda x: np.abs(x.filled_area - catphan_size))[0]
        out[f"{i}.best"] = np.array([best.label, best.filled_area, *best.centroid, *best.bbox], dtype=float)
np.savez_compressed(sys.argv[2], **out)
