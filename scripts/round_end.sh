#!/bin/bash
# One gpurun call that refreshes every piece of evidence the round is judged on:
#   full GPU suite (no -x) -> smoke() -> default bench line (all configs + CPU baselines) -> rocprofv3 kernel stats + two PMC
#   passes of the bench -> VALU counters of the Gaussian kernels.   gpurun --timeout 1500 -- 'bash scripts/round_end.sh r02g'
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/summary.txt
timeout 600 python bench.py > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"])
for k, c in d.get("configs", {}).items():
    print(k, c["value"], c["unit"], c["ms_per_pass"], c.get("cpu_baseline", {}).get("pool"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("pool"))
PY
bash scripts/profile_bench.sh $TAG > $OUT/profile_bench.log 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary.txt
cp gpurun_out/prof_$TAG/summary.json $OUT/rocprofv3_summary.json
grep '"metric"' gpurun_out/prof_$TAG/bench_trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
bash scripts/pmc_write_pipeline.sh > $OUT/pmc_pipeline_traffic.txt 2>&1
cp gpurun_out/pmc_pipe/pmc_traffic.json $OUT/pmc_traffic.json
cat $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
bash scripts/pmc_gauss.sh 2>&1 | grep -v "^[EW]2026" > $OUT/pmc_gauss.txt
tail -20 $OUT/pmc_gauss.txt | tee -a $OUT/summary.txt
