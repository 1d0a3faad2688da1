#!/bin/bash
# First GPU call of a new round (run through gpurun; ~4-5 GPU-minutes): everything that round 1 left unmeasured, in one
# call because every gpurun call costs ~35-40 s of overhead.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash scripts/round2_first_call.sh'
# Results land in gpurun_out/r2_first/ ; copy what matters into profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2_first
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp

# 1. the whole parity suite (includes the tests that never ran on hardware in round 1: the tail of test_gpu_parity.py)
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt

# 2. the run-based union-find start (PL_CCL_RUNS=1): parity first, then A/B timing of configs #3-#5
PL_CCL_RUNS=1 timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
  -k "label or fill or clear or catphan or features or fields or wl or canny or phantom or translation_equivariance" \
  > $OUT/pytest_ccl_runs.log 2>&1
echo "ccl-runs pytest rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/pytest_ccl_runs.log | tee -a $OUT/summary.txt
timeout 60 python scripts/time_configs.py $OUT/configs_ccl_px.jsonl 3 > $OUT/configs_px.log 2>&1
PL_CCL_RUNS=1 timeout 60 python scripts/time_configs.py $OUT/configs_ccl_runs.jsonl 3 > $OUT/configs_runs.log 2>&1
echo "--- per-pixel union-find" | tee -a $OUT/summary.txt; cat $OUT/configs_ccl_px.jsonl | tee -a $OUT/summary.txt
echo "--- run-based union-find" | tee -a $OUT/summary.txt; cat $OUT/configs_ccl_runs.jsonl | tee -a $OUT/summary.txt

# 2b. orientation timings of the "next"-row paths
timeout 120 python scripts/time_next_rows.py $OUT/next_rows.jsonl 3 > $OUT/next_rows.log 2>&1
echo "--- next-row paths" | tee -a $OUT/summary.txt; cat $OUT/next_rows.jsonl | tee -a $OUT/summary.txt

# 2c. single-rank RCCL: init -> all_gather_into_tensor -> barrier inside the bench step
PL_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 200 \
  python bench.py --no-cpu-baseline > $OUT/bench_force_dist.log 2>&1
echo "force-dist rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/bench_force_dist.log | cut -c1-300 | tee -a $OUT/summary.txt

# 3. the bench line (default), and the float64 Gaussian for reference
timeout 200 python bench.py 2>&1 | tail -1 > $OUT/bench.json
PL_GAUSS_PK=0 timeout 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_f64_gauss.json
# axis-0 packed kernel bounded for 4 workgroups per CU (128 VGPRs): parity of the Gaussian tests, then the bench
PL_GAUSS_V_OCC=4 timeout 120 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "gaussian or filters" > $OUT/pytest_occ4.log 2>&1
echo "occ4 pytest rc=$?" | tee -a $OUT/summary.txt
PL_GAUSS_V_OCC=4 timeout 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_v_occ4.json
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r2_first")
for name in ("bench.json", "bench_f64_gauss.json", "bench_v_occ4.json"):
    try:
        d = json.load(open(os.path.join(out, name)))
        print(name, d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["stage_ms"])
    except Exception as exc:
        print(name, "unreadable:", exc)
PY
