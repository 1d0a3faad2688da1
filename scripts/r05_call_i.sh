#!/bin/bash
# Round 5, call i: suite; picket-fence pass A/B on one box: the column-mean pass walking the batch last frame first (product)
# against first frame first (build/variants/lib_colfwd.so, -DPL_COLMEAN_REVERSE=0); kernel stats of both; WL pass; bench line.
TAG=${1:-r05i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
stats() {
  python - "$1" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    if "at::native" not in r["Name"]:
        print(f'   {r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
}
{ for lib in "" build/variants/lib_colfwd.so "" build/variants/lib_colfwd.so; do
    echo "== run_pf_pass.py 256 20, library: ${lib:-product (last frame first)}"
    for i in 1 2; do PYLINAC_HIP_LIB=${lib:-$GRAFT_REPO_ROOT/pylinac_amd/libpylinac_hip.so} timeout 300 python scripts/run_pf_pass.py 256 20 | tail -1; done
  done
  for lib in "" build/variants/lib_colfwd.so; do
    echo "== kernel stats, library: ${lib:-product (last frame first)}"
    rm -rf /tmp/prof_pf; PYLINAC_HIP_LIB=${lib:-$GRAFT_REPO_ROOT/pylinac_amd/libpylinac_hip.so} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf -o p -- python scripts/run_pf_pass.py 256 4 > /dev/null 2>&1
    stats /tmp/prof_pf
  done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/pf_colmean_order_ab.txt
{ for i in 1 2 3; do timeout 300 python scripts/run_wl_pass.py 1250 5 2>&1 | tail -1; done
  rm -rf /tmp/prof_wl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o p -- python scripts/run_wl_pass.py 512 4 > /dev/null 2>&1
  stats /tmp/prof_wl
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
