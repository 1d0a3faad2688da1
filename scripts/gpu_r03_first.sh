#!/bin/bash
# Round 3, first call: full GPU suite (incl. the bench-size parity tests), the MFMA / VALU overlap microbenchmark with cycle
# counters, and the default bench line (stated config sizes + parity samples).   gpurun --timeout 1200 -- 'bash scripts/gpu_r03_first.sh'
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=8 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu_full.log | head -20 | tee -a $OUT/summary.txt
timeout 120 scripts/ubench/mfma_valu_settle > $OUT/ubench_mfma_valu_settle.txt 2>&1
tail -60 $OUT/ubench_mfma_valu_settle.txt | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
tail -5 $OUT/bench_stderr.log | tee -a $OUT/summary.txt
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample"))
for k, c in d.get("configs", {}).items():
    print(k, c["value"], c["unit"], c["ms_per_pass"], c["parity_sample"]["ok"], c.get("cpu_baseline", {}).get("value"), c.get("cpu_baseline", {}).get("pool"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("pool"))
PY
