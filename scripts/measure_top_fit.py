"""would the closed-form bounded maximum of field_data's degree-2 fit reproduce the reference's "top" index?
The reference calls scipy.optimize.minimize (L-BFGS-B, pylinac/core/profile.py:1583-1593) and its 20 frozen fixtures pin THAT answer
to 1e-4 (tests_basic/core/test_profile.py:2546-2688).  Build container only (emulated kernels):
    python scripts/measure_top_fit.py > profiles/r06_top_fit_closed_form_vs_lbfgsb.txt"""
import sys, warnings
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
warnings.filterwarnings("ignore")
from pylinac_amd import profile as P
calls = []
real = P._bounded_top
def spy(fit_params, x0, lo, hi):
    r = real(fit_params, x0, lo, hi)
    a, b, c = fit_params
    if a < 0:
        xv = min(max(-b / (2 * a), lo), hi)
    else:
        xv = lo if (a*lo*lo + b*lo + c) >= (a*hi*hi + b*hi + c) else hi
    calls.append((r[0], xv, r[1], a*xv*xv + b*xv + c, a, hi - lo))
    return r
P._bounded_top = spy
from emu_backend import emulated_device
import importlib.util
spec = importlib.util.spec_from_file_location("fx", "/root/reference/tests_basic/core/profile_regression_fixtures.py")
g = np.load("/root/repo/tests/golden/single_profile.npz", allow_pickle=False)
print([k for k in g.files][:12])
with emulated_device():
    for i in range(20):
        p = P.SingleProfile(g[f"fx{i}.y"], interpolation=None)
        p.field_data()
        p2 = P.SingleProfile(g[f"fx{i}.y"], interpolation="Linear")
        p2.field_data()
c = np.array(calls)
d = np.abs(c[:, 0] - c[:, 1])
print("calls", len(c), "max |L-BFGS-B - vertex| in index:", d.max(), "median", np.median(d), "over 1e-4:", int((d > 1e-4).sum()), "over 1e-9:", int((d > 1e-9).sum()))
print("max |value diff|", np.abs(c[:, 2] - c[:, 3]).max())
for row in c[np.argsort(-d)[:5]]:
    print("  lbfgsb %.9f vertex %.9f  a=%.3e  window %.1f" % (row[0], row[1], row[4], row[5]))
