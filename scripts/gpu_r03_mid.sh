#!/bin/bash
# Round 3, mid-session check: full GPU suite, the bench line, per-kernel stats of the configuration paths.
TAG=${1:-r03m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
grep -E "FAILED|Error" $OUT/pytest_gpu_full.log | head -20 | tee -a $OUT/summary.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained"))
for k, c in d.get("configs", {}).items():
    print(k, c["value"], c["unit"], c["ms_per_pass"], c["parity_sample"]["ok"])
PY
tail -5 $OUT/bench_stderr.log | tee -a $OUT/summary.txt
bash scripts/profile_configs.sh pf ctp wln wl > $OUT/configs_kernel_stats.txt 2>&1
cut -c1-200 $OUT/configs_kernel_stats.txt | grep -v "at::native" | head -90
