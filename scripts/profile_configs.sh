#!/bin/bash
# rocprofv3 kernel-trace stats of the config #4 / #5 per-image paths (development aid).
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/run_cfg.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
which = sys.argv[1]
dev = torch.device("cuda:0")
if which == "wl":
    from pylinac_amd import winston_lutz
    from pylinac_amd.synthetic import wl_frames
    fr = torch.from_numpy(wl_frames(512)).to(dev)
    fn = lambda: winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
elif which == "wln":
    from pylinac_amd import winston_lutz
    from pylinac_amd.synthetic import wl_frames
    fr = torch.from_numpy(wl_frames(256, noise_sigma=0.001)).to(dev)
    fn = lambda: winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
elif which == "pf":
    from pylinac_amd import picketfence
    from pylinac_amd.synthetic import pf_frames
    fr = pf_frames(256, device=dev)
    fn = lambda: picketfence.analyze_batch(fr, 1 / 0.390625, num_pickets=10)
elif which == "ctp25":                      # config #5 at the bench's size: 25 volumes = one GPU's share of 200
    from pylinac_amd import ct
    from pylinac_amd.synthetic import catphan_volume
    vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(25)]).to(dev)
    fn = lambda: ct.ctp528_batch(vols, 0.5)
elif which == "ctp":
    from pylinac_amd import ct
    from pylinac_amd.synthetic import catphan_volume
    vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(4)]).to(dev)
    fn = lambda: ct.ctp528_batch(vols, 0.5)
else:
    from pylinac_amd import ct
    hs, mmpp = 512, 0.5
    yy, xx = torch.meshgrid(torch.arange(hs, device=dev), torch.arange(hs, device=dev), indexing="ij")
    r = torch.hypot((yy - hs * 0.49).double(), (xx - hs * 0.51).double()) * mmpp
    sl = torch.full((hs, hs), -1000.0, device=dev, dtype=torch.float64)
    sl[r < 100] = 60.0
    sl[(r < 100) & (((yy // 9) + (xx // 7)) % 2 == 0)] = 95.0
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    sh = torch.randint(-36, 36, (200, 2), generator=g)
    slices = torch.stack([torch.roll(sl.to(torch.int16), (int(a), int(b)), dims=(0, 1)) for a, b in sh])
    fn = lambda: ct.phantom_roi_batch(slices, mmpp)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): fn()
torch.cuda.synchronize()
print(which, "ms per pass", (time.perf_counter() - t0) / 3 * 1e3, flush=True)
PY
for which in ${@:-wl wln pf ctp25}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$which -o p -- python /tmp/run_cfg.py $which > $OUT/$which.log 2>&1
  grep "ms per pass" $OUT/$which.log
  python - "$OUT/$which" <<'PY'
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:90]
        print(f'{name:92s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.3f} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
  find $OUT/$which -name "*kernel_trace.csv" -delete
done
