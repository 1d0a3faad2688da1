#!/bin/bash
# bench line (stated configs + parity samples) + gauss2d_mm attribution variants
TAG=${1:-r03b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_variants.sh > /dev/null 2>&1
cp gpurun_out/g2d_variants.txt $OUT/
cat $OUT/g2d_variants.txt | tee $OUT/summary.txt
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -rf -k "bench_line or without_host_taps or gaussian" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)" | tee -a $OUT/summary.txt
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
