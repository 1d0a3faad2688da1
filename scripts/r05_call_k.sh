#!/bin/bash
# Round 5, call k: suite on the tree with the three-tier BB sweep (first pass: three workgroups per CU), then the
# Winston-Lutz A/B of r05_call_j.sh against the previous commit's library, then the bench line.
TAG=${1:-r05k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
bash scripts/r05_call_j.sh ${TAG}_ab
cp $GRAFT_REPO_ROOT/gpurun_out/${TAG}_ab/wl_sweep_occupancy_ab.txt $OUT/
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained", {}).get("value"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("parity_sample", {}).get("ok"))
PY
