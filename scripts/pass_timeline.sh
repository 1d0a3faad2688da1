#!/bin/bash
# Development aid: the device's kernels and copies of ONE pass of a BASELINE configuration in launch order, with start offsets,
# durations and the idle time in front of each (rocprofv3 --kernel-trace --memory-copy-trace; the last pass follows a 50 ms sleep).
#   gpurun -- 'bash scripts/pass_timeline.sh r06t pf wl'      (pf: config #3, wl: config #4, ct: config #5, epid: the headline step)
TAG=${1:-timeline}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
cat > /tmp/run_tl.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
which = sys.argv[1]
dev = torch.device("cuda:0")
if which == "wl":
    from pylinac_amd import winston_lutz
    from pylinac_amd.synthetic import wl_frames
    fr = torch.from_numpy(wl_frames(1250)).to(dev)
    fn = lambda: winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
elif which == "pf":
    from pylinac_amd import picketfence
    from pylinac_amd.synthetic import pf_frames
    fr = pf_frames(512, device=dev)
    fn = lambda: picketfence.analyze_batch(fr, 1 / 0.390625, num_pickets=10)
elif which == "ct":
    from pylinac_amd import ct
    from pylinac_amd.synthetic import catphan_volume
    vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(25)]).to(dev)
    fn = lambda: ct.ctp528_batch(vols, 0.5)
else:
    from pylinac_amd.pipeline import EpidPipeline
    from pylinac_amd.synthetic import pf_frames
    fr = pf_frames(256, h=1024, w=1024, device=dev)
    pipe = EpidPipeline(256, 1024, 1024, dev)
    fn = lambda: pipe.run(fr).record()
for _ in range(4):
    fn(); torch.cuda.synchronize()
time.sleep(0.05)
t0 = time.perf_counter()
fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(which, "last pass: returned after %.1f us, device done after %.1f us" % ((t1 - t0) * 1e6, (time.perf_counter() - t0) * 1e6), flush=True)
PY
for which in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/raw_$which -o p -- python /tmp/run_tl.py $which > $OUT/$which.log 2>&1
  grep "last pass" $OUT/$which.log | tee $OUT/timeline_$which.txt
  python - $OUT/raw_$which <<'PY' | tee -a $OUT/timeline_$which.txt
import csv, glob, sys, re
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")[:70]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
gaps = [(ev[i][0] - max(e[1] for e in ev[:i]), i) for i in range(1, len(ev))]
i0 = max(gaps)[1]                       # the 50 ms sleep
t0 = ev[i0][0]
prev_end, busy = t0, 0
for s, e, n in ev[i0:]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  idle before {(s - prev_end) / 1e3:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
PY
  rm -rf $OUT/raw_$which
done
