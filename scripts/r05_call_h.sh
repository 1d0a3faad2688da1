#!/bin/bash
# Round 5, call h: suite; the small-batch sweep of the headline step and the kernel list of a 32-frame step (the window
# Otsu kernel's merge no longer waits for one L2 round trip per bin); the Winston-Lutz pass with the edge strips taken
# from the histogram's own stream; the bench line.
TAG=${1:-r05l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
for f in 256 64 32 8; do timeout 120 python bench.py --gpus 1 --frames $f --steps 30 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames', $f, 'ms/step', d['ms_per_step'], 'us/frame', round(d['ms_per_step']*1e3/$f, 3), d['roofline']['stage_ms'])"; done | tee $OUT/small_batch_sweep.txt
{ echo "== kernels of bench.py --frames 32 --steps 50"
  rm -rf /tmp/prof_sb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sb -o p -- python bench.py --gpus 1 --frames 32 --steps 50 --warmup 10 --no-cpu-baseline --no-configs > /dev/null 2>&1
  python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/prof_sb/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'   {r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
} 2>&1 | grep -v amdgpu.ids | tee $OUT/small_batch_kernels.txt
{ for mode in "" "noise"; do
    echo "== run_wl_pass.py 1250 5 $mode"
    for i in 1 2 3; do timeout 300 python scripts/run_wl_pass.py 1250 5 $mode 2>&1 | tail -1; done
  done
  echo "== kernel stats, 512 frames"
  rm -rf /tmp/prof_wl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o p -- python scripts/run_wl_pass.py 512 4 > /dev/null 2>&1
  python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/prof_wl/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:10]:
    print(f'   {r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
} 2>&1 | grep -v amdgpu.ids | tee $OUT/wl_pass.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained", {}).get("value"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("parity_sample", {}).get("ok"))
PY
