"""MEASURE how the reference orders exact ties in find_peaks (pylinac/core/profile.py:2616-2618:
``sorted(list(np.argsort(peak_props[peak_sort]))[::-1][:max_number])``) instead of asserting it.  Build container only
(needs /root/reference):   python scripts/measure_tie_order.py > profiles/r04_find_peaks_tie_order.txt

The device keeps, among equal keys, the LATER peaks (np.argsort(kind="stable")[::-1][:max_number]).  np.argsort's default kind
is introsort, which numpy >= 1.25 dispatches to x86-simd-sort (AVX-512 / AVX2 sorting networks) for float64: its order of equal
keys is neither stable nor a simple rule, and it changes with the CPU's SIMD level."""
import sys
import warnings

import numpy as np

sys.path.insert(0, ".")
warnings.filterwarnings("ignore")
from oracle import ref_loader  # noqa: E402

prof = ref_loader.ref("core.profile")
import numpy._core._multiarray_umath as mu  # noqa: E402

print("numpy", np.__version__, "| SIMD:", [f for f in ("AVX2", "AVX512F", "AVX512_SKX") if mu.__cpu_features__.get(f)])
rng = np.random.default_rng(0)
unstable = {}
for n in (2, 3, 4, 8, 15, 16, 17, 32, 100):
    bad = sum(not np.array_equal(np.argsort(a), np.argsort(a, kind="stable"))
              for a in (rng.integers(0, 4, n) * 0.25 for _ in range(400)))
    unstable[n] = bad
print("np.argsort(default) != np.argsort(kind='stable') on arrays with ties, of 400 trials per length:", unstable)


def make(hs):
    x = np.zeros(10 * len(hs) + 10)
    for i, hgt in enumerate(hs):
        c = 10 * i + 7
        x[c - 2:c + 3] = np.array([0.25, 0.5, 1.0, 0.5, 0.25]) * hgt
    return x


rng = np.random.default_rng(3)
tot = tie_cut = diff = 0
examples = []
for _ in range(3000):
    k = int(rng.integers(3, 25))
    hs = rng.choice([1.0, 1.5, 2.0], k)
    mx = int(rng.integers(1, k))
    idx, _p = prof.find_peaks(make(hs), max_number=mx, peak_sort="peak_heights")
    kept = sorted(((np.asarray(idx) - 7) // 10).tolist())
    stable = sorted(np.argsort(hs, kind="stable")[::-1][:mx].tolist())
    srt = np.sort(hs)[::-1]
    tot += 1
    tie_cut += bool(srt[mx - 1] == srt[mx])
    if kept != stable:
        diff += 1
        if len(examples) < 3:
            examples.append((hs.tolist(), mx, kept, stable))
print(f"reference find_peaks on {tot} profiles of 3-24 peaks with heights from {{1, 1.5, 2}}: {tie_cut} have an exact tie across the "
      f"max_number cut; the reference keeps a different set than the stable rule in {diff} of them")
for e in examples:
    print("   heights", e[0], "max_number", e[1], "reference kept", e[2], "stable rule", e[3])
all_equal = sum(sorted(((np.asarray(prof.find_peaks(make(np.ones(k)), max_number=mx)[0]) - 7) // 10).tolist()) == list(range(k - mx, k))
                for k in range(2, 41) for mx in range(1, k))
print(f"all peaks equal (k = 2..40, every max_number): the reference keeps the LAST max_number peaks in {all_equal} of "
      f"{sum(k - 1 for k in range(2, 41))} cases (= the stable rule)")
print("conclusion: with ties ACROSS the cut the reference's answer is the sorting network's, i.e. platform-defined; it is not")
print("reproducible on another numpy build or CPU, let alone on the device.  Peak keys of measured profiles are float64 sums /")
print("interpolations of noisy data: exact ties do not occur (none in any golden or synthetic batch of this repository).")
