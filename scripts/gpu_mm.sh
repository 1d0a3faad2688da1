#!/bin/bash
# MFMA Gaussian bring-up on the MI355X: operand-layout microbenchmark, the Gaussian parity tests, the bench line.
OUT=$GRAFT_REPO_ROOT/gpurun_out/mm
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -rf -k "gaussian or filters or pipeline or smoke or epid or median or otsu or full_size" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20 | tee -a $OUT/summary.txt
timeout 200 python bench.py --no-cpu-baseline --no-configs --steps 20 2>&1 | tail -1 > $OUT/bench.json
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "mm", "bench.json")))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])
PY
bash scripts/pmc_gauss.sh 2>&1 | grep -v "^[EW]2026" | tee -a $OUT/summary.txt
