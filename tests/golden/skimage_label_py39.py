"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): measure.label of every mask in
an .npz with connectivity 1 and 2 -> .npz.  Used only by make_golden.py in the build container."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from skimage import measure

d = np.load(sys.argv[1])
out = {}
for k in d.files:
    for conn in (1, 2):
        out[f"{k}.conn{conn}"] = measure.label(d[k], connectivity=conn).astype(np.int32)
np.savez_compressed(sys.argv[2], **out)
