"""Helper run under /opt/conda/bin/python3.9 (scikit-image 0.18.3 for draw.disk): the reference's OWN
pylinac.core.gamma.gamma_2d (pylinac/core/gamma.py:229-330).  Build container only."""
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

sys.path.insert(0, sys.argv[3])
from oracle import ref_loader as rl

rl._STUB_ROOTS = tuple(list(rl._STUB_ROOTS) + ["pydantic", "plotly", "tabulate", "tqdm"])
gm = rl.ref("core.gamma")
d = np.load(sys.argv[1], allow_pickle=True)
out = {}
for k in range(int(d["count"])):
    kw = d[f"kw{k}"].item()
    out[f"g{k}"] = gm.gamma_2d(reference=d[f"ref{k}"], evaluation=d[f"ev{k}"], **kw)
np.savez_compressed(sys.argv[2], **out)
