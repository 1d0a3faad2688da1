"""Golden vectors for the geometric 1-D gamma (pylinac/core/gamma.py:105-227) and ``PhysicalProfileMixin.gamma``
(pylinac/core/profile.py:822-874), produced by the reference's OWN functions: its known-answer inputs
(tests_basic/core/test_gamma.py:304-372) and dose-like profile pairs with shifted / scaled / coarser / reversed evaluation
samples.  Build container only:

    python tests/golden/make_gamma_geometric_golden.py        # -> tests/golden/gamma_geometric.npz
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402

warnings.filterwarnings("ignore")
gm = ref_loader.ref("core.gamma")
prof = ref_loader.ref("core.profile")


def beam(x, centre, width, pen, amp=100.0, base=2.0):
    return amp / (1 + np.exp(-(x - (centre - width / 2)) / pen)) / (1 + np.exp((x - (centre + width / 2)) / pen)) + base


rng = np.random.default_rng(17)
x = np.arange(120, dtype=float)
cases = [
    dict(reference=np.ones(5), evaluation=np.ones(5)),                                             # test_same_profile_is_0_gamma
    dict(reference=np.ones(5), evaluation=np.ones(5) * 1.01, dose_to_agreement=1),                 # test_gamma_perfectly_at_1
    dict(reference=np.ones(5), evaluation=np.ones(5) * 0.99, dose_to_agreement=1),
    dict(reference=np.ones(5), evaluation=np.ones(5) * 1.005, dose_to_agreement=1),                # test_gamma_half
    dict(reference=np.ones(5), evaluation=np.ones(5) * 1.03, dose_to_agreement=1, gamma_cap_value=2),   # capped
    dict(reference=beam(x, 60, 50, 2.5), evaluation=beam(x, 60.8, 50.5, 2.7) * 1.01),
    dict(reference=beam(x, 60, 50, 2.5), evaluation=beam(x, 61.5, 49, 2.5), dose_to_agreement=2, distance_to_agreement=2,
         dose_threshold=10, fill_value=0.0),
    dict(reference=beam(x, 60, 50, 2.5) + rng.normal(0, 0.3, x.size), evaluation=beam(x[::2], 60.4, 50, 2.5),
         reference_coordinates=x * 0.5, evaluation_coordinates=x[::2] * 0.5, distance_to_agreement=1.5, dose_to_agreement=3),
    dict(reference=beam(x, 60, 50, 2.5), evaluation=beam(x, 60.3, 50, 2.5)[::-1].copy(), evaluation_coordinates=x[::-1].copy(),
         distance_to_agreement=0.7),                                                                # decreasing coordinates
    dict(reference=(beam(x, 60, 50, 2.5) * 400).astype(np.uint16), evaluation=(beam(x, 59.5, 50, 2.5) * 395).astype(np.uint16),
         gamma_cap_value=5, dose_threshold=0),                                                      # integer doses
]
out = {"count": np.int64(len(cases))}
for k, c in enumerate(cases):
    kw = {}
    for name, v in c.items():
        if isinstance(v, np.ndarray):
            out[f"{name}{k}"] = v
        else:
            kw[name] = v
    out[f"kw{k}"] = np.array(json.dumps(kw))
    out[f"gamma{k}"] = gm.gamma_geometric(**c)

# PhysicalProfileMixin.gamma: FWXMProfilePhysical pairs (dpmm given; physical x-values without dpmm)
ref_v, ev_v = beam(x, 60, 50, 2.5), beam(x, 60.6, 50.4, 2.6) * 1.004
r1, e1 = prof.FWXMProfilePhysical(ref_v, dpmm=2.0), prof.FWXMProfilePhysical(ev_v, dpmm=2.0)
g1, rr, ee = r1.gamma(e1, dose_to_agreement=1, distance_to_agreement=1, return_profiles=True)
xs2 = np.linspace(-30, 29.5, 90)
r2 = prof.FWXMProfilePhysical(ref_v, dpmm=2.0)
e2 = prof.FWXMProfilePhysical(beam(xs2 * 2 + 60, 60.2, 50, 2.5), x_values=xs2 + 0.3, dpmm=None)
g2 = r2.gamma(e2, dose_to_agreement=2, distance_to_agreement=2, gamma_cap_value=3, dose_threshold=8, fill_value=-1.0)
out.update({"p.ref": ref_v, "p.ev": ev_v, "p.gamma1": g1, "p.ref_x1": rr.x_values, "p.ev_x1": ee.x_values, "p.xs2": xs2,
            "p.ev2": np.asarray(e2.values), "p.gamma2": g2})
np.savez_compressed(os.path.join(HERE, "gamma_geometric.npz"), **out)
for k in range(len(cases)):
    print(k, np.round(np.nanmax(out[f"gamma{k}"]), 4), np.round(np.nanmean(out[f"gamma{k}"]), 4))
print(np.nanmax(g1), np.nanmax(g2))
