"""Ad-hoc GPU smoke/parity/timing run used while bringing kernels up (not part of the test suite)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pylinac_amd import ops
from oracle import pylinac_oracle as o
from scipy import ndimage, signal

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def T(a): return torch.from_numpy(a).to(dev)
ok = True
def chk(name, cond):
    global ok
    print(("PASS " if cond else "FAIL ") + name, flush=True); ok &= bool(cond)

for shape in [(3, 64, 80), (2, 100, 1000), (1, 256, 512), (2, 7, 9)]:
    a = rng.integers(0, 65536, shape, dtype=np.uint16)
    for sigma in (1, 2, 5, 2.7):
        ref = np.stack([ndimage.gaussian_filter(f, sigma) for f in a])
        got = ops.gaussian_filter(T(a), sigma).cpu().numpy()
        chk(f"gaussian u16 {shape} s={sigma} maxdiff={np.abs(ref.astype(int)-got.astype(int)).max()}", np.array_equal(ref, got))
    for size in (3, 2, 5):
        ref = np.stack([ndimage.median_filter(f, size=size) for f in a])
        got = ops.median_filter(T(a), size).cpu().numpy()
        chk(f"median u16 {shape} size={size}", np.array_equal(ref, got))
a = rng.integers(0, 65536, (2, 128, 256), dtype=np.uint16)
af = (a / 7.0)
for dt in (np.float64, np.float32):
    x = af.astype(dt)
    ref = np.stack([ndimage.gaussian_filter(f, 2) for f in x]); got = ops.gaussian_filter(T(x), 2).cpu().numpy()
    chk(f"gaussian {dt.__name__}", np.array_equal(ref, got))
    ref = np.stack([ndimage.median_filter(f, size=3) for f in x]); got = ops.median_filter(T(x), 3).cpu().numpy()
    chk(f"median3 {dt.__name__}", np.array_equal(ref, got))
ai = (a.astype(np.int32) - 32768).astype(np.int16)
chk("gaussian i16", np.array_equal(np.stack([ndimage.gaussian_filter(f, 2) for f in ai]), ops.gaussian_filter(T(ai), 2).cpu().numpy()))
# elementwise
mn, mx = ops.minmax(T(a)); chk("minmax", np.array_equal(mn.cpu().numpy(), a.min(axis=(1,2))) and np.array_equal(mx.cpu().numpy(), a.max(axis=(1,2))))
chk("ground", np.array_equal(ops.ground(T(a)).cpu().numpy(), np.stack([o.ground(f) for f in a])))
chk("normalize", np.array_equal(ops.normalize(T(a)).cpu().numpy(), np.stack([o.normalize(f) for f in a])))
chk("invert", np.array_equal(ops.invert(T(a)).cpu().numpy(), np.stack([o.invert(f) for f in a])))
chk("threshold hi", np.array_equal(ops.threshold(T(a), 30000).cpu().numpy(), o.threshold(a, 30000)))
chk("threshold lo", np.array_equal(ops.threshold(T(a), 30000, "low").cpu().numpy(), o.threshold(a, 30000, "low")))
chk("as_binary", np.array_equal(ops.as_binary(T(a), 30000).cpu().numpy(), o.as_binary(a, 30000)))
# hist / otsu / percentile
g = np.stack([ndimage.gaussian_filter(f, 3) for f in a])
h = ops.histogram16(T(g)).cpu().numpy().view(np.uint32)
chk("hist16", all(np.array_equal(h[i], np.bincount(g[i].ravel(), minlength=65536)) for i in range(len(g))))
chk("otsu", np.array_equal(ops.threshold_otsu(T(g)).cpu().numpy(), np.array([o.threshold_otsu(f) for f in g])))
chk("otsu i16", np.array_equal(ops.threshold_otsu(T(ai)).cpu().numpy(), np.array([o.threshold_otsu(f) for f in ai])))
q = [0.5, 5, 50, 99.5, 99.9, 0, 100]
chk("percentile", np.array_equal(ops.percentile(T(a), q).numpy(), np.stack([np.percentile(f, q) for f in a])))
for ax in (0, 1):
    for op in ("mean", "sum", "max", "min"):
        ref = getattr(np, op)(a, axis=ax + 1).astype(np.float64)
        chk(f"reduce {op} axis{ax}", np.array_equal(ops.reduce_axis(T(a), ax, op).cpu().numpy(), ref))
thr = ops.threshold_otsu(T(g)); th, cs = ops.threshold_colsum_u16(T(g), thr)
refth = np.stack([o.threshold(f, int(t)) for f, t in zip(g, thr.cpu().numpy())]).astype(np.uint16)
chk("threshold_colsum", np.array_equal(th.cpu().numpy(), refth) and np.array_equal(cs.cpu().numpy(), refth.sum(axis=1, dtype=np.int64)))
# peaks
npk = 0
for trial in range(60):
    L = int(rng.integers(5, 3000)); x = np.abs(rng.normal(size=L).cumsum()) if trial % 3 else rng.integers(0, 9, L).astype(float)
    kws = [dict(), dict(threshold=0.3, peak_separation=0.05), dict(threshold=0.5, peak_separation=0.02, peak_sort="peak_heights", required_prominence=0.1*np.ptp(x), max_number=3),
           dict(search_region=(0.2, 0.8), max_number=2), dict(fwxm_height=0.3, max_number=1)]
    kw = kws[trial % 5]
    if trial % 3 == 0 and kw.get("peak_separation", 0): kw = dict(kw, peak_separation=0)   # ties + distance: undefined order
    i1, p1 = o.find_peaks(x, **kw)
    i2, p2 = ops.find_peaks_batch(T(x), **kw).to_host(0)
    good = np.array_equal(i1, i2) and all(np.array_equal(p1[k], p2[k]) for k in p1)
    npk += len(i1)
    if not good: chk(f"find_peaks trial {trial} {kw}", False)
chk(f"find_peaks 60 trials ({npk} peaks)", True)

# timing on config #2 shapes
n = 64
fr = torch.from_numpy(rng.integers(0, 65536, (n, 1024, 1024), dtype=np.uint16)).to(dev)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
out = torch.empty_like(fr); tmp = torch.empty_like(fr)
t = timeit(lambda: ops.gaussian_filter(fr, 5, out=out, tmp=tmp)); print(f"gaussian s=5: {t/n*1e6:.2f} us/frame")
out2 = torch.empty_like(fr)
t = timeit(lambda: ops.median_filter(out, 3, out=out2)); print(f"median3: {t/n*1e6:.2f} us/frame")
hist = torch.empty((n, 65536), dtype=torch.int32, device=dev)
t = timeit(lambda: ops.histogram16(out2, out=hist)); print(f"hist16: {t/n*1e6:.2f} us/frame")
t = timeit(lambda: ops.otsu_from_hist(hist, torch.uint16)); print(f"otsu: {t/n*1e6:.2f} us/frame")
thr = ops.otsu_from_hist(hist, torch.uint16)[0]
t = timeit(lambda: ops.threshold_colsum_u16(out2, thr, out=out)); print(f"thr+colsum: {t/n*1e6:.2f} us/frame")
prof = ops.reduce_axis(out, 0, "mean")
t = timeit(lambda: ops.find_peaks_batch(prof, fwxm_height=0.5, max_number=1)); print(f"find_peaks: {t/n*1e6:.2f} us/frame")
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
