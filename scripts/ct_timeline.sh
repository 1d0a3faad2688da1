#!/bin/bash
# Development aid: the kernels of ONE config #5 pass in launch order with their start offsets, durations and the idle time in
# front of each (rocprofv3 --kernel-trace timestamps of the last of four passes of scripts/profile_configs.sh's ctp25 driver).
#   gpurun -- 'bash scripts/ct_timeline.sh r06r'
TAG=${1:-timeline}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX"; exit 7; }
cat > /tmp/run_ct_tl.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from pylinac_amd import ct
from pylinac_amd.synthetic import catphan_volume
dev = torch.device("cuda:0")
vols = torch.stack([torch.from_numpy(catphan_volume(4000 + v)) for v in range(25)]).to(dev)
for _ in range(3):
    ct.ctp528_batch(vols, 0.5)
    torch.cuda.synchronize()
t0 = time.perf_counter()
ct.ctp528_batch(vols, 0.5)
torch.cuda.synchronize()
print("last pass wall ms", (time.perf_counter() - t0) * 1e3, flush=True)
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/raw -o p -- python /tmp/run_ct_tl.py > $OUT/run.log 2>&1
grep "last pass" $OUT/run.log | tee $OUT/timeline.txt
python - $OUT/raw <<'PY' | tee -a $OUT/timeline.txt
import csv, glob, sys, re
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")[:70]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the last pass starts at the last edge kernel's init
starts = [i for i, e in enumerate(ev) if e[2].startswith("e32_init_kernel")]
i0 = starts[-1]
t0 = ev[i0][0]
prev_end = t0
busy = 0
for s, e, n in ev[i0:]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  idle before {(s - prev_end) / 1e3:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
PY
rm -rf $OUT/raw
