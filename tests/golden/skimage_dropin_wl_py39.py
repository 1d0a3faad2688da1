"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): the reference's OWN ``WLBaseImage.analyze()`` -- the whole
method, pylinac/winston_lutz.py:669-762 -- on array images that borrow the WLBaseImage / LinacDicomImage members it touches
unchanged (DicomImage.cax, pylinac/core/image.py:1549-1580; SURVEY.md section 8c: the constructors need pydicom, the analysis does not).  Frames = those of tests/golden/wl.npz.
The frozen results are what tests/test_dropin_reference.py compares the SAME method, running over pylinac_amd's image / metric
classes, against.  Build container only:

    /opt/conda/bin/python3.9 tests/golden/skimage_dropin_wl_py39.py tests/golden/dropin_wl.npz /root/repo
"""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[2])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
from skimage.measure._regionprops import RegionProperties

RegionProperties.area_filled = property(lambda self: self.filled_area)       # the reference uses the >=0.19 names
RegionProperties.area_bbox = property(lambda self: self.bbox_area)
image = rl.ref("core.image")
geometry = rl.ref("core.geometry")
wl = rl.ref("winston_lutz")

BORROWED = ("analyze", "_clean_edges", "find_field_centroids", "find_field_matches", "find_bb_centroids", "find_bb_matches",
            "nominal_bb_position", "_calculate_bb_tolerance", "field_to_bb_distances", "epid_to_bb_distances", "epid")


def wl_image_class(wl, image, base):
    """an array image class with WLBaseImage's per-image members (and LinacDicomImage.cax) bound unchanged"""
    class W(base):
        detection_conditions = wl.WinstonLutz2D.detection_conditions      # winston_lutz.py:1144-1150
        gantry_angle, collimator_angle, couch_angle, sad = 0.0, 0.0, 0.0, 1000.0

    for name in BORROWED:
        setattr(W, name, wl.WLBaseImage.__dict__[name])
    W.cax = image.DicomImage.__dict__["cax"]                              # no translation tags -> the image centre
    return W


def iso_arrangement():
    """BBArrangement.ISO (winston_lutz.py:111-120) as plain objects (BBConfig is a pydantic model; pydantic is a stub here)"""
    import types

    return (types.SimpleNamespace(name="Iso", offset_left_mm=0, offset_up_mm=0, offset_in_mm=0, bb_size_mm=5, rad_size_mm=20),)


def run(W, frame, pixel_mm, gantry=0.0, couch=0.0, **kw):
    img = W(frame.copy(), dpi=25.4 / pixel_mm)
    img.gantry_angle, img.couch_angle = gantry, couch
    img.analyze(bb_arrangement=iso_arrangement(), **kw)
    m = img.arrangement_matches["Iso"]
    v = m.bb_field_vector_mm
    return np.array([m.field.x, m.field.y, m.bb.x, m.bb.y, v.x, v.y, v.z, m.bb_field_distance_mm, m.bb_epid_distance_mm,
                     m.field_epid_distance_mm, img.shape[0], img.shape[1]], dtype=float)


if __name__ == "__main__":
    g = np.load(sys.argv[2] + "/tests/golden/wl.npz")
    W = wl_image_class(wl, image, image.ArrayImage)
    pixel = float(g["pixel_mm"])
    cases = [(0, {}, 0.0, 0.0), (6, {}, 0.0, 0.0), (7, {}, 0.0, 0.0), (8, dict(bb_proximity_mm=30), 45.0, 10.0),
             (2, dict(shift_vector=geometry.Vector(x=0.4, y=-0.3, z=0.2), snap_tolerance=1), 90.0, 0.0)]
    rows = [run(W, g["frames"][k], pixel, gantry, couch, **kw) for k, kw, gantry, couch in cases]
    np.savez_compressed(sys.argv[1], frame_index=np.array([c[0] for c in cases]), gantry=np.array([c[2] for c in cases]),
                        couch=np.array([c[3] for c in cases]), record=np.stack(rows),
                        columns=np.array("field_x field_y bb_x bb_y bb_field_vx bb_field_vy bb_field_vz bb_field_mm bb_epid_mm "
                                         "field_epid_mm rows cols".split()))
    print(np.round(np.stack(rows), 4))
