"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN spatial-resolution chain per slice
(pylinac/ct.py:1511-1580: combine_surrounding_slices(+-3, "max") -> CollapsedCircleProfile(20 radii, 2x sampling, start
angle pi, ccw) -> filter(0.001, "gaussian") -> ground -> per line-pair region find_peaks / find_valleys -> MTF) on a
synthetic CatPhan volume (config #5 recipe), with the phantom centre from the reference's find_phantom_axis fits.
Build container only:

    /opt/conda/bin/python3.9 tests/golden/skimage_ctp528_py39.py tests/golden/ctp528.npz /root/repo
"""
import importlib.util
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
_spec = importlib.util.spec_from_file_location("pl_synthetic", sys.argv[2] + "/pylinac_amd/synthetic.py")   # no package init
_syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_syn)

MMPP, SIZE, N = 0.65, 384, 16
vol, truth = _syn.catphan_volume(4000, N, SIZE, MMPP, return_truth=True)
stack = [image.load(s.copy()) for s in vol]
meta = types.SimpleNamespace(SliceThickness=2.5, PixelSpacing=[MMPP, MMPP])


class Stack(list):
    metadata = meta


dstack = Stack(stack)
dstack.slice_spacing = 2.5
cp = types.SimpleNamespace(dicom_stack=dstack, clear_borders=True, x_adjustment=0, y_adjustment=0,
                           catphan_size=np.pi * 101 ** 2 / MMPP ** 2, mm_per_pixel=MMPP, clip_in_localization=False,
                           _phantom_center_func=None, num_images=N)
fit_zx, fit_zy = ct.CatPhanBase.find_phantom_axis(cp)
out = {"volume": vol, "mmpp": np.float64(MMPP), "fit_zx": np.asarray(fit_zx.coeffs, float), "fit_zy": np.asarray(fit_zy.coeffs, float),
       "resolution_slice": np.int64(truth["resolution_slice"])}
slices = list(range(3, N - 3))
profiles, rmtf, nregions, maxs_all, mins_all = [], [], [], [], []
for s in slices:
    m = object.__new__(ct.CTP528CP504)
    m.origin_slice, m._offset, m.slice_spacing = s, 0, 2.5          # slice_num = origin_slice + round(offset / spacing)
    m._phantom_center_func = (fit_zx, fit_zy)
    m.scaling_factor, m.mm_per_pixel, m.catphan_roll, m.roi_size_factor = 1, MMPP, 0.0, 1
    m.image = image.load(ct.combine_surrounding_slices(dstack, s, slices_plusminus=3, mode="max"))
    prof = m.circle_profile
    profiles.append(np.asarray(prof.values, dtype=float))
    row = np.full(8, np.nan)
    try:
        mtf = m.mtf
        vals = list(mtf.norm_mtfs.values())            # sorted by spacing = region order
        row[: len(vals)] = vals
        nregions.append(len(mtf.maximums))
        mx, mn = np.full(8, np.nan), np.full(8, np.nan)
        mx[: len(mtf.maximums)] = mtf.maximums
        mn[: len(mtf.minimums)] = mtf.minimums
    except ValueError:
        nregions.append(0)
        mx, mn = np.full(8, np.nan), np.full(8, np.nan)
    rmtf.append(row)
    maxs_all.append(mx)
    mins_all.append(mn)
out["slices"] = np.array(slices)
out["profiles"] = np.stack(profiles)
out["rmtf"] = np.stack(rmtf)
out["nregions"] = np.array(nregions)
out["maxs"], out["mins"] = np.stack(maxs_all), np.stack(mins_all)
np.savez_compressed(sys.argv[1], **out)
print(out["fit_zx"], out["fit_zy"], out["profiles"].shape)
print(out["nregions"])
print(np.round(out["rmtf"], 3))
