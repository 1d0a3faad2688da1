"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN RectangleROI
(pylinac/core/roi.py:481-704: vertices via EuclideanTransform, pixels_flat via skimage.draw.polygon) on a CT-like
slice, plus raw skimage.draw.polygon pixel lists for random polygons.  Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from skimage import draw, transform

# pylinac's Rectangle.vertices (core/geometry.py:692-704) hands matrix_transform a transform OBJECT; newer scikit-image
# takes np.asarray() of it (= its 3x3 .params), 0.18.3 wants the matrix itself -- pass .params, same arithmetic.
_mt = transform.matrix_transform
transform.matrix_transform = lambda coords, m: _mt(coords, getattr(m, "params", m))

sys.path.insert(0, sys.argv[3])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
roi = rl.ref("core.roi")
geo = rl.ref("core.geometry")

d = np.load(sys.argv[1])
out = {}
# width, height, cx, cy, rotation
rects = np.array([[20, 12, 256, 256, 0], [20.5, 12.3, 180.4, 300.7, 0], [30, 10, 330.5, 199.5, 30],
                  [8, 40, 100.2, 120.9, -47.5], [25, 25, 256, 256, 45], [60, 40, 20, 20, 10], [2, 2, 300.5, 300.5, 0],
                  [90, 70, 255.5, 255.5, 90], [40, 6, 480, 500, -12.25], [3.7, 2.2, 50.1, 60.9, 133]], dtype=float)
out["rects"] = rects
for name in ("slice_i16", "slice_f32"):
    arr = d[name].astype(np.float64) if name == "slice_f32" else d[name]
    rows, verts = [], []
    for w, h, cx, cy, rot in rects:
        m = roi.RectangleROI(arr, width=float(w), height=float(h), center=geo.Point(float(cx), float(cy)), rotation=float(rot))
        flat = m.pixels_flat
        rows.append([flat.size, m.mean, m.std, m.min, m.max, float(np.median(flat)), m.pixel_value])
        verts.append([[v.x, v.y] for v in m.vertices])
    out["stats_" + name[6:]] = np.array(rows, dtype=float)
    out["vertices"] = np.array(verts, dtype=float)
# unrotated pixel_array (roi.py:664-681): its own rounding of the corners
pa = []
for w, h, cx, cy, rot in rects:
    if rot == 0:
        m = roi.RectangleROI(d["slice_i16"], width=float(w), height=float(h), center=geo.Point(float(cx), float(cy)))
        a = m.pixel_array
        pa.append([a.shape[0], a.shape[1], a.mean(), a.std()])
out["pixel_array"] = np.array(pa, dtype=float)
fc = roi.RectangleROI.from_phantom_center(d["slice_i16"], width=14.5, height=9.25, angle=-60.0, dist_from_center=80.5,
                                          phantom_center=geo.Point(250.3, 260.7), rotation=22.5)
out["from_center"] = np.array([fc.center.x, fc.center.y, fc.pixels_flat.size, fc.mean, fc.std, fc.min, fc.max])

# raw skimage.draw.polygon: integer and float vertices, concave shapes, polygons leaving the image, with / without shape
rng = np.random.default_rng(17)
shape = (64, 80)
polys, rr_all, cc_all, offs = [], [], [], [0]
nverts = []
for k in range(40):
    nv = int(rng.integers(3, 8))
    if k % 4 == 0:
        r, c = rng.integers(-5, 70, nv).astype(float), rng.integers(-5, 86, nv).astype(float)
    elif k % 4 == 1:
        r, c = rng.uniform(5, 60, nv), rng.uniform(5, 75, nv)
    elif k % 4 == 2:
        r, c = np.round(rng.uniform(-10, 75, nv) * 2) / 2, np.round(rng.uniform(-10, 90, nv) * 2) / 2
    else:
        ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
        r, c = 30 + 20 * np.sin(ang), 40 + 25 * np.cos(ang)
    rr, cc = draw.polygon(r, c, shape=shape)
    p = np.full((8, 2), np.nan)
    p[:nv, 0], p[:nv, 1] = r, c
    polys.append(p)
    nverts.append(nv)
    rr_all.append(rr)
    cc_all.append(cc)
    offs.append(offs[-1] + len(rr))
out["poly_shape"] = np.array(shape)
out["poly_vertices"] = np.array(polys)
out["poly_nverts"] = np.array(nverts)
out["poly_rr"] = np.concatenate(rr_all).astype(np.int64)
out["poly_cc"] = np.concatenate(cc_all).astype(np.int64)
out["poly_offsets"] = np.array(offs, dtype=np.int64)
np.savez_compressed(sys.argv[2], **out)
