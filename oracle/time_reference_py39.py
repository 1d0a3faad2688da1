"""TEST / BENCH INFRASTRUCTURE ONLY.  Helper of time_reference.py, run under /opt/conda/bin/python3.9 (scikit-image 0.18.3):
the reference's OWN Winston-Lutz per-image sequence (pylinac/winston_lutz.py:668-806) and its OWN CatPhan phantom ROI +
CTP528CP504 chain (pylinac/ct.py:2398-2445, 1511-1580) timed on the synthetic inputs of configs #4 / #5."""
import json
import sys
import time
import types
import warnings

warnings.filterwarnings("ignore")
import numpy as np

inp, outp, root = sys.argv[1:4]
sys.path.insert(0, root)
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm", "matplotlib", "PIL", "webbrowser"])
from skimage.measure._regionprops import RegionProperties

RegionProperties.area_filled = property(lambda self: self.filled_area)
RegionProperties.area_bbox = property(lambda self: self.bbox_area)
image = rl.ref("core.image")
wl = rl.ref("winston_lutz")
ct = rl.ref("ct")
data = np.load(inp)


def median_time(fn, warmup=3, repeats=5):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


class W(image.ArrayImage):
    detection_conditions = wl.WinstonLutz2D.detection_conditions


for name in ("_clean_edges", "find_field_centroids", "find_bb_centroids", "_calculate_bb_tolerance"):
    setattr(W, name, getattr(wl.WLBaseImage, name))


def run_wl(frames):
    for f in frames:
        img = W(f.copy(), dpi=25.4 / 0.336)
        img.check_inversion_by_histogram(percentiles=(0.01, 50, 99.99))
        img._clean_edges()
        img.ground()
        img.normalize()
        img.find_field_centroids(is_open_field=False)
        img.find_bb_centroids(bb_diameter_mm=5.0, low_density=False)


def run_ct(vol, slices, mmpp=0.5):
    """per slice: the phantom ROI (find_phantom_axis's per-slice work) + the CTP528 chain about the fitted centre"""
    n = len(vol)
    stack = [image.load(s.copy()) for s in vol]

    class Stack(list):
        metadata = types.SimpleNamespace(SliceThickness=2.5, PixelSpacing=[mmpp, mmpp])

    dstack = Stack(stack)
    dstack.slice_spacing = 2.5
    cp = types.SimpleNamespace(dicom_stack=dstack, clear_borders=True, x_adjustment=0, y_adjustment=0,
                               catphan_size=np.pi * 101 ** 2 / mmpp ** 2, mm_per_pixel=mmpp, clip_in_localization=False,
                               _phantom_center_func=None, num_images=n)
    fit_zx, fit_zy = ct.CatPhanBase.find_phantom_axis(cp)          # phantom ROI of EVERY slice of the stack
    for s in slices:
        m = object.__new__(ct.CTP528CP504)
        m.origin_slice, m._offset, m.slice_spacing = s, 0, 2.5
        m._phantom_center_func = (fit_zx, fit_zy)
        m.scaling_factor, m.mm_per_pixel, m.catphan_roll, m.roi_size_factor = 1, mmpp, 0.0, 1
        m.image = image.load(ct.combine_surrounding_slices(dstack, s, slices_plusminus=3, mode="max"))
        try:
            _ = m.mtf
        except ValueError:
            pass


out = {}
for key, tag in (("#4", "wl"), ("#4n", "wln")):
    fr = data[tag]
    dt = median_time(lambda: run_wl(fr))
    out[key] = {"reference": {"value": round(len(fr) / dt, 3), "unit": "frames/s", "cores": 1,
                              "sample": f"{len(fr)} units, median repeat {dt:.3f} s",
                              "interpreter": "python 3.9 / scipy 1.7.1 / scikit-image 0.18.3"},
                "what": "WLBaseImage.analyze()'s per-image sequence on 1024 x 1024 uint16 frames" + (" with RandomNoiseLayer(0.001)" if tag == "wln" else " (noise-free)")}
vol = data["ct"]
sl = list(range(3, len(vol) - 3))
dt = median_time(lambda: run_ct(vol, sl), warmup=1, repeats=3)
out["#5"] = {"reference": {"value": round(len(vol) / dt, 3), "unit": "slices/s", "cores": 1,
                           "sample": f"{len(vol)}-slice 512 x 512 volume (phantom ROI of every slice + CTP528 on {len(sl)} of them), 1 warm-up + 3 repeats, median {dt:.3f} s",
                           "interpreter": "python 3.9 / scipy 1.7.1 / scikit-image 0.18.3"},
             "what": "CatPhanBase.find_phantom_axis + CTP528CP504 per slice"}
json.dump(out, open(outp, "w"))
