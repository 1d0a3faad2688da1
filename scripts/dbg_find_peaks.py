"""Development aid: tests/test_gpu_parity.py::test_find_peaks_vs_oracle_random with every trial reported."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import pylinac_oracle as o
from pylinac_amd import ops, _lib
dev = torch.device("cuda:0")
for rep in range(3):
    rng = np.random.default_rng(17)
    bad = []
    for trial in range(80):
        L = int(rng.integers(3, 6000))
        x = rng.integers(0, 7, L).astype(float) if trial % 4 == 0 else np.abs(rng.normal(size=L).cumsum())
        kws = [dict(), dict(threshold=0.3, peak_separation=0.05),
               dict(threshold=0.5, peak_separation=0.02, peak_sort="peak_heights", required_prominence=0.1 * np.ptp(x), max_number=3),
               dict(search_region=(0.2, 0.8), max_number=2), dict(fwxm_height=0.3, max_number=1),
               dict(threshold=0.2, peak_separation=3, peak_sort="widths", max_number=4)]
        kw = dict(kws[trial % len(kws)])
        if trial % 4 == 0:
            kw.pop("peak_separation", None); kw.pop("max_number", None)
        i1, p1 = o.find_peaks(x, **kw)
        try:
            i2, p2 = ops.find_peaks_batch(torch.from_numpy(x).to(dev), **kw).to_host(0)
            ok = np.array_equal(i1, i2) and all(np.array_equal(p1[k], p2[k]) for k in p1)
            if not ok: bad.append((trial, L, kw, "mismatch", len(i1), len(i2)))
        except _lib.PylinacHipError as e:
            bad.append((trial, L, kw, str(e)[:60], len(i1)))
    print("rep", rep, "bad", bad, flush=True)
