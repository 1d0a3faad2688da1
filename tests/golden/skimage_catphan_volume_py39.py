"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN CatPhanBase.find_phantom_axis and
find_origin_slice (pylinac/ct.py:2398-2508) driven over a synthetic CatPhan-like volume through a stand-in analyzer object
that carries only the attributes those methods read.  Build container only."""
import sys
import types
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[2])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm", "matplotlib", "PIL", "webbrowser"])
ct = rl.ref("ct")
image = rl.ref("core.image")


def volume(n, size, seed, hu_slices, tilt=(0.12, -0.07), thickness=2.5):
    """n slices: air, a 200 mm cylinder whose centre drifts linearly with z (a tilted phantom), HU-module slices with
    high / low inserts on the 58 mm circle, uniform slices elsewhere, two empty slices, a couch bar, noise"""
    rng = np.random.default_rng(seed)
    mmpp = 250.0 / size
    y, x = np.mgrid[0:size, 0:size].astype(float)
    out = []
    for z in range(n):
        img = np.full((size, size), -1000.0)
        if z not in (0, n - 1):                       # first / last slice: no phantom
            cy, cx = size / 2 + 3.0 + tilt[0] * z, size / 2 - 2.0 + tilt[1] * z
            r = np.hypot(y - cy, x - cx) * mmpp
            img[r < 100] = 90.0
            if z in hu_slices:
                for k in range(8):
                    a = k * np.pi / 4 + 0.1
                    iy, ix = cy + 58 / mmpp * np.sin(a), cx + 58 / mmpp * np.cos(a)
                    img[np.hypot(y - iy, x - ix) * mmpp < 6.5] = [-1000, 340, -200, 950, -100, 120, -1000, 990][k]
            elif z % 5 == 0:                          # a resolution-like module: fine bar pattern, no big HU swings
                ring = (np.abs(r - 48) < 3) & (((np.arctan2(y - cy, x - cx) * 20).astype(int) % 2) == 0)
                img[ring] = 300.0
        img[int(size * 0.975):int(size * 0.995), :] = 200.0
        img += rng.normal(0, 10, img.shape)
        out.append(np.round(img))
    return np.stack(out).astype(np.int16), mmpp


out = {}
for name, (n, size, seed, hu) in {"a": (24, 256, 3, (9, 10, 11, 12, 13)), "b": (18, 200, 4, (4, 5, 6))}.items():
    vol, mmpp = volume(n, size, seed, hu)
    stack = [image.load(s.copy()) for s in vol]
    meta = types.SimpleNamespace(SliceThickness=2.5, PixelSpacing=[mmpp, mmpp])

    class Stack(list):
        metadata = meta

    dstack = Stack(stack)
    dstack.slice_spacing = 2.5
    cp = types.SimpleNamespace(roll_slice_offset=0, air_bubble_radius_mm=7, dicom_stack=dstack, clear_borders=True, x_adjustment=0, y_adjustment=0,
                               catphan_size=np.pi * 101 ** 2 / mmpp ** 2, mm_per_pixel=mmpp, clip_in_localization=False,
                               _phantom_center_func=None, num_images=n, localization_radius=59,
                               hu_origin_slice_variance=400, _is_within_image_extent=lambda k: 0 <= k < n)
    fit_zx, fit_zy = ct.CatPhanBase.find_phantom_axis(cp)
    cp._phantom_center_func = (fit_zx, fit_zy)
    origin = ct.CatPhanBase.find_origin_slice(cp)
    cp.origin_slice = origin
    cp._is_right_area = types.MethodType(ct.CatPhanBase._is_right_area, cp)
    cp._is_right_eccentricity = types.MethodType(ct.CatPhanBase._is_right_eccentricity, cp)
    out[f"{name}.roll"] = np.float64(ct.CatPhanBase.find_phantom_roll(cp))
    s0 = ct.Slice(cp, origin, clear_borders=True)
    _, regs, _ = ct.get_regions(s0)
    out[f"{name}.roll_regions"] = np.array([[r.area, r.filled_area, r.eccentricity, r.centroid[0], r.centroid[1]] for r in regs], dtype=float)
    # per-slice intermediate values for diagnosis
    in_view, cen = [], []
    for idx, img in enumerate(dstack):
        s = ct.Slice(cp, slice_num=idx, clear_borders=True, original_image=img)
        ok = s.is_phantom_in_view()
        in_view.append(ok)
        cen.append(s.phantom_roi.centroid if ok else (np.nan, np.nan))
    out[f"{name}.volume"], out[f"{name}.mmpp"] = vol, np.float64(mmpp)
    out[f"{name}.fit_zx"], out[f"{name}.fit_zy"] = np.asarray(fit_zx.coeffs, float), np.asarray(fit_zy.coeffs, float)
    out[f"{name}.origin"] = np.int64(origin)
    out[f"{name}.in_view"] = np.array(in_view)
    out[f"{name}.centroids"] = np.array(cen, dtype=float)
    print(name, fit_zx.coeffs, fit_zy.coeffs, origin, int(np.sum(in_view)), out[f"{name}.roll"], len(regs))
np.savez_compressed(sys.argv[1], **out)
