#!/bin/bash
# Development loop on the MI355X: [pytest -k expr] + the bench line + (optional) PMC view of the Gaussian kernels.
#   gpurun -- 'bash scripts/gpu_check.sh "gaussian or filters or pipeline" pmc'
OUT=$GRAFT_REPO_ROOT/gpurun_out/check
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ -n "$1" ]; then
  timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -rf -k "$1" > $OUT/pytest.log 2>&1
  echo "pytest rc=$? $(tail -1 $OUT/pytest.log)" | tee -a $OUT/summary.txt
  grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20 | tee -a $OUT/summary.txt
fi
timeout 200 python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 > $OUT/bench.json
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "check", "bench.json")))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])
PY
if [ "$2" = "pmc" ]; then bash scripts/pmc_gauss.sh 2>&1 | grep -v "^[EW]2026" | tee -a $OUT/summary.txt; fi
