"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): ``skimage.transform.rotate(array, angle, mode="edge")``
exactly as BaseImage.rotate calls it (pylinac/core/image.py:780-783), for the dtypes an image array can have, together with
the 3x3 inverse map skimage handed to ``warp`` (captured, so that the kernel can be checked bit for bit on the same map and
the host's construction of the map separately).
Build container only:  /opt/conda/bin/python3.9 tests/golden/skimage_rotate_py39.py tests/golden/rotate.npz"""
import sys
import numpy as np
from skimage.transform import _warps, rotate

captured = {}
_warp = _warps.warp


def spy(image, tform, **kw):
    captured["m"] = np.array(tform.params, dtype=np.float64)
    return _warp(image, tform, **kw)


_warps.warp = spy
rng = np.random.default_rng(11)
yy, xx = np.mgrid[0:37, 0:52]
base = 1000 + 30000 * np.exp(-((yy - 15.0) ** 2 / 120 + (xx - 30.0) ** 2 / 200)) + rng.normal(0, 200, yy.shape)
base[6:9, 4:20] += 9000                      # a bar, so that a wrong rotation sense or centre shows
arrays = {
    "u16": np.clip(base, 0, 65535).astype(np.uint16),
    "i16": (np.clip(base, 0, 65535) - 20000).astype(np.int16),
    "u8": (np.clip(base, 0, 65535) / 257).astype(np.uint8),
    "i32": (base * 3).astype(np.int32),
    "f64": base / 7.0,
    "f32": (base / 7.0).astype(np.float32),
    "f64_even": (base / 7.0)[:36, :48].copy(),
    "bool": base > 9000,
}
angles = [33.3, 90.0, 0.0, -12.5, 270.25, 45.0]
FEW = {"u8", "i32", "bool"}           # the remaining integer conversions: two angles are enough
out = {"names": np.array(list(arrays)), "angles": np.array(angles)}
for name, a in arrays.items():
    out[name + ".in"] = a
    for k, ang in enumerate(angles[:2] if name in FEW else angles):
        r = rotate(a, ang, mode="edge")
        out[f"{name}.{k}.out"] = r
        out[f"{name}.{k}.m"] = captured["m"]
np.savez_compressed(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out))
