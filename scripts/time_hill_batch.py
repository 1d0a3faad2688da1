"""Time profile.single_profile_hill_batch (INFLECTION_HILL for a batch of profiles, no per-profile host call)
against the per-profile mirror (scipy's curve_fit on the host) on synthetic open-field profiles.
    python scripts/time_hill_batch.py [n_profiles] [detectors]"""
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import next_row_checks as checks  # noqa: E402
from pylinac_amd import profile  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
length = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
profs = torch.from_numpy(checks.beam_profiles(n, length)).to(dev)
for norm in ("Beam center", "Max"):
    profile.single_profile_hill_batch(profs, normalization_method=norm)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        res = profile.single_profile_hill_batch(profs, normalization_method=norm)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    info = res.info.cpu().numpy()
    print(f"{n} profiles x {length} detectors (x10 resampled), normalisation {norm!r}: {best * 1e3:.2f} ms "
          f"({n / best:,.0f} profiles/s); converged {((info >= 1) & (info <= 4)).all(axis=1).sum()} of {n}; "
          f"function evaluations per fit: mean {res.nfev.float().mean().item():.0f}, max {res.nfev.max().item()}")
host = profs[:32].cpu().numpy()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t0 = time.perf_counter()
    for row in host:
        profile.SingleProfile(row.copy(), edge_detection_method=profile.Edge.INFLECTION_HILL).inflection_data()
    per = (time.perf_counter() - t0) / len(host)
print(f"per-profile mirror (device kernels + scipy curve_fit on the host): {per * 1e3:.2f} ms per profile")
