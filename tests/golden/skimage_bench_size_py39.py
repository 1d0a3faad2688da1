"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3) by make_bench_size_golden.py: the reference's OWN
per-image Winston-Lutz sequence (WLBaseImage.analyze, pylinac/winston_lutz.py:668-806) and its OWN CTP528CP504 chain
(pylinac/ct.py:1511-1580, axis fits pylinac/ct.py:2398-2445) on inputs handed over as .npz -- the BASELINE-size batches
bench.py times (configs #4 and #5).  Build container only:

    /opt/conda/bin/python3.9 tests/golden/skimage_bench_size_py39.py wl|ct in.npz out.npz /root/repo
"""
import sys
import types
import warnings

warnings.filterwarnings("ignore")
import numpy as np

kind, inp, outp, root = sys.argv[1:5]
sys.path.insert(0, root)
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm", "matplotlib", "PIL", "webbrowser"])
from skimage.measure._regionprops import RegionProperties

RegionProperties.area_filled = property(lambda self: self.filled_area)       # the reference uses the >=0.19 names
RegionProperties.area_bbox = property(lambda self: self.bbox_area)
image = rl.ref("core.image")
data = np.load(inp)
out = {}

if kind == "wl":
    wl = rl.ref("winston_lutz")

    class W(image.ArrayImage):
        """an array image that borrows WLBaseImage's per-image methods unchanged"""
        detection_conditions = wl.WinstonLutz2D.detection_conditions      # winston_lutz.py:1144-1150

    for name in ("_clean_edges", "find_field_centroids", "find_bb_centroids", "_calculate_bb_tolerance"):
        setattr(W, name, getattr(wl.WLBaseImage, name))
    pixel_mm, bb_mm = float(data["pixel_mm"]), float(data["bb_mm"])
    rec, shapes, inverted = [], [], []
    for f in data["frames"]:
        img = W(f.copy(), dpi=25.4 / pixel_mm)
        inverted.append(bool(img.check_inversion_by_histogram(percentiles=(0.01, 50, 99.99))))
        img._clean_edges()
        shapes.append(img.array.shape)
        img.ground()
        img.normalize()
        fld = img.find_field_centroids(is_open_field=False)[0]
        try:
            bbs = img.find_bb_centroids(bb_diameter_mm=bb_mm, low_density=False)
            bb = (bbs[0].x, bbs[0].y, len(bbs))
        except ValueError:
            bb = (np.nan, np.nan, 0)
        rec.append([fld.x, fld.y, bb[0], bb[1], bb[2]])
    out["record"] = np.array(rec, dtype=float)
    out["shape_after_clean"] = np.array(shapes, dtype=np.int64)
    out["inverted"] = np.array(inverted)
elif kind == "ct":
    ct = rl.ref("ct")
    vol, mmpp = data["volume"], float(data["mmpp"])
    n = len(vol)
    stack = [image.load(s.copy()) for s in vol]

    class Stack(list):
        metadata = types.SimpleNamespace(SliceThickness=2.5, PixelSpacing=[mmpp, mmpp])

    dstack = Stack(stack)
    dstack.slice_spacing = 2.5
    cp = types.SimpleNamespace(dicom_stack=dstack, clear_borders=True, x_adjustment=0, y_adjustment=0,
                               catphan_size=np.pi * 101 ** 2 / mmpp ** 2, mm_per_pixel=mmpp, clip_in_localization=False,
                               _phantom_center_func=None, num_images=n)
    fit_zx, fit_zy = ct.CatPhanBase.find_phantom_axis(cp)
    out["fit_zx"], out["fit_zy"] = np.asarray(fit_zx.coeffs, float), np.asarray(fit_zy.coeffs, float)
    # slices 0 .. 2: combine_surrounding_slices indexes dicomstack[-3 .. -1] there, which WRAPS to the end of the stack;
    # slices n - 3 .. n - 1 raise IndexError in the reference (checked below) and have no golden rows
    slices = list(range(0, n - 3))
    for s in range(n - 3, n):
        try:
            ct.combine_surrounding_slices(dstack, s, slices_plusminus=3, mode="max")
            raise SystemExit("the reference combined a window that passes the end of the stack")
        except IndexError:
            pass
    profiles, rmtf, nregions, maxs_all, mins_all = [], [], [], [], []
    for s in slices:
        m = object.__new__(ct.CTP528CP504)
        m.origin_slice, m._offset, m.slice_spacing = s, 0, 2.5
        m._phantom_center_func = (fit_zx, fit_zy)
        m.scaling_factor, m.mm_per_pixel, m.catphan_roll, m.roi_size_factor = 1, mmpp, 0.0, 1
        m.image = image.load(ct.combine_surrounding_slices(dstack, s, slices_plusminus=3, mode="max"))
        profiles.append(np.asarray(m.circle_profile.values, dtype=float))
        row, mx, mn = np.full(8, np.nan), np.full(8, np.nan), np.full(8, np.nan)
        try:
            mtf = m.mtf
            vals = list(mtf.norm_mtfs.values())
            row[: len(vals)] = vals
            nregions.append(len(mtf.maximums))
            mx[: len(mtf.maximums)] = mtf.maximums
            mn[: len(mtf.minimums)] = mtf.minimums
        except ValueError:
            nregions.append(0)
        rmtf.append(row)
        maxs_all.append(mx)
        mins_all.append(mn)
    out["slices"] = np.array(slices)
    out["profiles"] = np.stack(profiles)
    out["rmtf"], out["nregions"] = np.stack(rmtf), np.array(nregions)
    out["maxs"], out["mins"] = np.stack(maxs_all), np.stack(mins_all)
np.savez(outp, **out)
