#!/bin/bash
# WRITE_SIZE (and FETCH_SIZE) of the EPID pipeline's kernels from a minimal driver (no input-generation kernels of torch
# beyond one batch): fallback when the PMC passes of the full bench hang.  Output: gpurun_out/pmc_pipe/summary.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_pipe; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/run_pipe.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from pylinac_amd.pipeline import EpidPipeline
from pylinac_amd.synthetic import epid_open_field_frames
dev = torch.device("cuda:0")
fr = epid_open_field_frames(256, 1024, 1024, device=dev)
pipe = EpidPipeline(256, 1024, 1024, dev)
for _ in range(3):
    pipe.run(fr)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python /tmp/run_pipe.py > $OUT/$c.log 2>&1
  echo "$c rc=$?"
done
python3 - <<'PY' | tee gpurun_out/pmc_pipe/summary.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_pipe/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).split("(")[0].replace("void ", "")
        if any(k in n for k in ("gauss2d", "otsu16_window", "median3_threshold", "median3_oct", "find_peaks", "hist16", "otsu_kernel")):
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
for n, d in sorted(acc.items()):
    print(n, {k: round(sum(v) / len(v), 1) for k, v in d.items()}, "(KiB per launch; FETCH counts 64 B per 128-B request: double it)")
import json
stage_of = {"gauss2d_mm": "gauss2d", "otsu16_window_kernel": "median3_otsu16", "median3_threshold_colsum_kernel": "median3_threshold_colsum",
            "find_peaks_kernel": "find_peaks"}
out = {"_comment": "HBM bytes per launch (256 frames 1024x1024 u16) = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc "
                   "passes over a minimal driver of the pipeline (scripts/pmc_write_pipeline.sh; FETCH doubled per MI355X_MICROARCH.md section HBM)"}
for n, d in acc.items():
    for key, stage in stage_of.items():
        if n.startswith(key) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            out[stage] = int(round((2 * f + w) * 1024))
json.dump(out, open("gpurun_out/pmc_pipe/pmc_traffic.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +1M -delete
