#!/bin/bash
# Round 5, after the evidence run: the -m gpu suite on the final tree (pl_hill_fit_ex's settled-fit bar), and the PMC traffic
# passes of the pipeline again (r05z's WRITE_SIZE pass ran into its 80 s time-out) with 200 s each.
TAG=${1:-r05d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
timeout 500 bash scripts/pmc_write_pipeline.sh > $OUT/pmc_pipeline_traffic.txt 2>&1
cp gpurun_out/pmc_pipe/pmc_traffic.json $OUT/pmc_traffic.json
cat $OUT/pmc_pipeline_traffic.txt $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
timeout 300 python scripts/time_hill_batch.py 4096 200 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/hill_batch.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("parity_sample", {}).get("ok"))
PY
