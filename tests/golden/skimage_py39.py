"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3): threshold_otsu of every array
in an .npz -> JSON.  Used only by make_golden.py in the build container."""
import json
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
from skimage.filters import threshold_otsu

d = np.load(sys.argv[1])
json.dump({k: int(threshold_otsu(d[k])) for k in d.files}, open(sys.argv[2], "w"))
