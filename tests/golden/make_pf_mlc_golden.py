"""Golden vectors for picket-fence analysis on the OTHER leaf banks ("no HD_MILLENNIUM / AGILITY / HALCYON case
anywhere"), produced by the reference's OWN PicketFence.analyze() (pylinac/picketfence.py:636-845) with ``mlc=`` one of its MLC
enum members, through the stub loader.  Build container only:

    python tests/golden/make_pf_mlc_golden.py        # -> tests/golden/picketfence_mlc.npz

Cases: HD_MILLENNIUM (14 x 5 + 32 x 2.5 + 14 x 5 mm leaves: 6-pixel and 13-pixel windows in one frame), AGILITY (80 x 5 mm),
HALCYON_DISTAL (28 x 10 mm: the tallest windows), BMOD (40 x 4 mm), each on its own synthetic UP_DOWN frame; AGILITY also
LEFT_RIGHT.
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


CASES = [("hd", "HD_MILLENNIUM", 420, 540, 0.39, 2101, False), ("agility", "AGILITY", 400, 520, 0.5, 2102, False),
         ("halcyon", "HALCYON_DISTAL", 440, 500, 0.6, 2103, False), ("bmod", "BMOD", 380, 500, 0.42, 2104, False),
         ("agility_lr", "AGILITY", 400, 520, 0.5, 2105, True)]
out = {"names": np.array([c[0] for c in CASES]), "mlcs": np.array([c[1] for c in CASES]),
       "transposed": np.array([c[6] for c in CASES])}
for name, mlc, hh, ww, pixel, seed, transpose in CASES:
    raw = pf_frame(hh, ww, pixel, seed, n_pickets=7, spacing_mm=20.0, gap_mm=2.5)
    if transpose:
        raw = np.ascontiguousarray(raw.T)
    dpmm = 1 / pixel
    im = PFImg(raw.copy(), dpi=dpmm * 25.4, sid=1000)
    im.crop(pixels=int(round(3 * im.dpmm)))             # picketfence.py:214-215
    cropped = np.ascontiguousarray(im.array)
    im.ground()
    im.normalize()                                       # picketfence.py:322-323
    pf = pfm.PicketFence(None, mlc=getattr(pfm.MLC, mlc))   # skips image loading (picketfence.py:315), keeps the bank
    pf.image = im
    pf.analyze(orientation="Left-Right" if transpose else "Up-Down", nominal_gap_mm=2.5)
    out[f"{name}.cropped"] = cropped
    out[f"{name}.dpmm"] = np.float64(im.dpmm)
    out[f"{name}.meas"] = np.array([[m.leaf_num, m.picket_num, m._approximate_idx] + list(m.position) for m in pf.mlc_meas])
    out[f"{name}.spacing"] = np.float64(pf.mlc_meas[0]._spacing)
    out[f"{name}.max_error"] = np.float64(pf.max_error)
    heights = sorted({int(m._image_window.shape[1 if transpose else 0]) for m in pf.mlc_meas})
    print(name, mlc, cropped.shape, len(pf.mlc_meas), "measurements; window heights", heights, "max error", round(pf.max_error, 4))
np.savez_compressed(os.path.join(HERE, "picketfence_mlc.npz"), **out)
