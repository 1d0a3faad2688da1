"""Golden vectors for the picket-fence orientation / separate-leaves cases (VERDICT r3 "missing" 2), produced by the reference's
OWN PicketFence.analyze() (pylinac/picketfence.py:636-845, MLCValue.get_peak_positions :1605-1628) through the stub loader,
real numpy / scipy underneath.  Build container only:

    python tests/golden/make_pf_orient_golden.py        # -> tests/golden/picketfence_orient.npz

Cases: a LEFT_RIGHT frame (the UP_DOWN generator's frame transposed, non-square so that the two axes cannot be confused),
the same frame with separate_leaves=True, and an UP_DOWN frame with separate_leaves=True.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref_loader  # noqa: E402
from make_golden import pf_frame  # noqa: E402

warnings.filterwarnings("ignore")
image = ref_loader.ref("core.image")
pfm = ref_loader.ref("picketfence")


class PFImg(image.ArrayImage):          # what PFDicomImage adds to the array image (picketfence.py:204-260)
    _central_axis = None

    def adjust_for_sag(self, sag, orientation):
        pass


out = {}
cases = [("lr", 400, 520, 0.78125, 2010, True, False), ("lr_sep", 400, 520, 0.78125, 2010, True, True),
         ("ud_sep", 384, 512, 0.8, 2011, False, True)]
for name, hh, ww, pixel, seed, transpose, separate in cases:
    raw = pf_frame(hh, ww, pixel, seed)
    if transpose:
        raw = np.ascontiguousarray(raw.T)
    dpmm = 1 / pixel
    im = PFImg(raw.copy(), dpi=dpmm * 25.4, sid=1000)
    im.crop(pixels=int(round(3 * im.dpmm)))             # picketfence.py:214-215
    cropped = np.ascontiguousarray(im.array)
    im.ground()
    im.normalize()                                       # picketfence.py:322-323
    pf = pfm.PicketFence(None)                           # skips image loading (picketfence.py:315)
    pf.image = im
    pf.analyze(orientation="Left-Right" if transpose else "Up-Down", separate_leaves=separate, nominal_gap_mm=2)
    if name != "lr_sep":                                 # (the same frame as "lr")
        out[f"{name}.cropped"] = cropped
    out[f"{name}.dpmm"] = np.float64(im.dpmm)
    out[f"{name}.meas"] = np.array([[m.leaf_num, m.picket_num, m._approximate_idx] + list(m.position) for m in pf.mlc_meas])
    out[f"{name}.spacing"] = np.float64(pf.mlc_meas[0]._spacing)
    print(name, cropped.shape, len(pf.mlc_meas), "measurements, positions per window:", len(pf.mlc_meas[0].position))
np.savez_compressed(os.path.join(HERE, "picketfence_orient.npz"), **out)
