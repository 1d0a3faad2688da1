#!/bin/bash
# A/B of the Gaussian kernels in the stopwatch harness + Gaussian parity group + short bench
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in $(ls scripts/ubench/g2d_v* | sort -V); do echo $b; timeout 60 $b; timeout 60 $b 256 1; done 2>&1 | tee $OUT/g2d_variants.txt
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -rf -k "gaussian or pipeline or filters or headline or full_size" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20 | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 20 2>$OUT/bench_stderr.log | tail -1 > $OUT/bench.json
python - $OUT/bench.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"))
PY
