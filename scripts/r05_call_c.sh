#!/bin/bash
# Round 5, third GPU call: suite, then Winston-Lutz A/B on one box (field CAX reading every frame whole / only the tiles the
# histogram pass's maxima allow; clean and noisy frames), the picket-fence pass, kernel stats of both, the bench line.
TAG=${1:-r05c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
{ for mode in "notiles" "" "notiles noise" "noise"; do
    echo "== run_wl_pass.py 1250 5 $mode"
    for i in 1 2; do timeout 300 python scripts/run_wl_pass.py 1250 5 $mode 2>&1 | tail -1; done
  done
  for mode in "notiles" ""; do
    echo "== kernel stats, 512 frames, $mode"
    rm -rf /tmp/prof_wl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o p -- python scripts/run_wl_pass.py 512 4 $mode > /dev/null 2>&1
    python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/prof_wl/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'   {r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
  done; } 2>&1 | grep -v amdgpu.ids | tee $OUT/wl_tile_maxima_ab.txt
{ for i in 1 2; do timeout 300 python scripts/run_pf_pass.py 256 20 | tail -1; done
  rm -rf /tmp/prof_pf; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf -o p -- python scripts/run_pf_pass.py 256 4 > /dev/null 2>&1
  python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/prof_pf/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    if "at::native" not in r["Name"]:
        print(f'   {r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
} 2>&1 | grep -v amdgpu.ids | tee $OUT/pf_pass.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained", {}).get("value"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("parity_sample", {}).get("ok"))
PY
