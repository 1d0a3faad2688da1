"""Helper run under /opt/conda/bin/python3.9: scikit-image 0.18.3's own feature.canny on float64 test images with
the parameters pylinac passes (pylinac/planar_imaging.py:198, 577-583, 1978).  Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from skimage import feature

d = np.load(sys.argv[1], allow_pickle=True)
out = {}
for k in range(int(d["count"])):
    kw = d[f"kw{k}"].item()
    out[f"edges{k}"] = feature.canny(d[f"img{k}"], **kw)
np.savez_compressed(sys.argv[2], **out)
