#!/bin/bash
# copy the judged summaries of a round_end run from gpurun_out/<tag>/ into profiles/<tag>_* and stamp the PMC traffic file
# with the commit the passes ran on:   scripts/stash_evidence.sh r03z
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=gpurun_out/$TAG
for f in summary.txt pytest_gpu_full.log bench_line_full.json bench_line_under_rocprof.json bench_three_runs.txt rocprofv3_summary.txt \
         pmc_pipeline_traffic.txt pmc_sq_pipeline_kernels.txt configs_kernel_stats.txt dvfs_ramp.txt pmc_traffic.json \
         pmc_sq_ct_kernels.txt pmc_sq_pf_kernels.txt wide_range.txt; do
  [ -s $SRC/$f ] && cp $SRC/$f profiles/${TAG}_$f
done
python3 - $TAG <<'PY'
import json, subprocess, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/{tag}/pmc_traffic.json"))
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
d["_measured_at"] = f"commit {commit} (run {tag}: profiles/{tag}_pmc_epid.txt)"
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
print(d)
PY
ls profiles | grep $TAG
