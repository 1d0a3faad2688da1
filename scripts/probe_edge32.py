import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pylinac_amd import ct, ops
from pylinac_amd.synthetic import catphan_volume
dev = torch.device("cuda:0")
vol = catphan_volume(seed=4000, n_slices=80)
x = torch.from_numpy(np.ascontiguousarray(vol[list(range(0, 80, 8))])).to(dev)
spans = ct._disk_spans_on_device(512, 512, 0.5, dev)
e64, rawmax64, lo64, hi64 = ops.edge_plane(x, 1, spans=spans, dtype=torch.float64)
p32, rawmax, lo, hi, status, B = ops.edge_plane32(x, 1, spans=spans)
want32 = e64.to(torch.float32)
d = (p32.view(torch.int32).to(torch.int64) - want32.view(torch.int32).to(torch.int64))
print("bracket", B, "max |d|", int(d.abs().max()), "mean |d|", float(d.abs().float().mean()), "hist of |d| > 8:", int((d.abs() > 8).sum()), "status", status.tolist())
n, h, w = x.shape
thr64, _ = ops.edge_otsu(e64, lo64, hi64, spans=spans, scale=0.8)
tvs = dict(thr=thr64, high=e64.reshape(n, -1).sort(dim=1).values[:, -(h * w) // 7].contiguous(),
           med32=want32.reshape(n, -1).sort(dim=1).values[:, h * w // 2].to(torch.float64).contiguous(),
           zero=torch.zeros_like(thr64), neg=torch.full_like(thr64, -1.0))
for name, tv in tvs.items():
    for br in (B, 64):
        q = ops.edge_regions(p32, x, 1, tv.contiguous(), 0, False, 64, return_mask=True, want_table=False, bracket=br)
        want = (e64 > tv[:, None, None]).to(torch.uint8)
        bad = (q["mask"] != want)
        print(name, "bracket", br, "mismatches", int(bad.sum()), "status", q["status"].tolist()[:3])
        if int(bad.sum()):
            idx = bad.nonzero()[:5]
            for i, r, c in idx.tolist():
                print("   slice", i, "r", r, "c", c, "e64", float(e64[i, r, c]), "p32", float(p32[i, r, c]), "t", float(tv[i]), "got", int(q["mask"][i, r, c]), "d", int(d[i, r, c]))
