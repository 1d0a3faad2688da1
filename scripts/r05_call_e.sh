#!/bin/bash
# Round 5: (1) HBM traffic of the three pipeline kernels on the final build through scripts/pmc_kernels.sh (FETCH_SIZE and
# WRITE_SIZE in separate passes, each under a time-out) over a driver with few framework launches -- pmc_write_pipeline.sh's
# WRITE_SIZE pass ran into its time-out twice this round; (2) the stopwatch bounds of gauss2d_mm (scripts/r05_g2d_bounds.sh).
TAG=${1:-r05e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 300 bash scripts/r05_g2d_bounds.sh 2>&1 | tee $OUT/gauss2d_bounds.txt
timeout 120 python scripts/run_epid_pass.py 256 5 | tail -1 | tee $OUT/epid_pass.txt
P=$GRAFT_REPO_ROOT/gpurun_out/pmc_epid; rm -rf $P; mkdir -p $P
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/$c -o p -- python scripts/run_epid_pass.py 256 3 > $P/$c.log 2>&1
  echo "$c rc=$?" | tee -a $OUT/pmc_epid.txt
done
python - <<'PY' | tee -a $OUT/pmc_epid.txt
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_epid/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).split("(")[0].replace("void ", "")
        if any(k in n for k in ("gauss2d", "otsu16_window", "median3_threshold", "find_peaks")):
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
stage_of = {"gauss2d_mm": "gauss2d", "otsu16_window_kernel": "median3_otsu16", "median3_threshold_colsum_kernel": "median3_threshold_colsum",
            "find_peaks_kernel": "find_peaks"}
out = {"_comment": "HBM bytes per launch (256 frames 1024x1024 u16) = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc "
                   "passes over scripts/run_epid_pass.py (scripts/r05_call_e.sh; FETCH doubled per MI355X_MICROARCH.md section HBM)"}
for n, d in sorted(acc.items()):
    med = {k: sorted(v)[len(v) // 2] for k, v in d.items()}
    print(n, {k: round(v, 1) for k, v in med.items()}, "KiB per launch (median)")
    for key, stage in stage_of.items():
        if n.startswith(key) and "FETCH_SIZE" in med and "WRITE_SIZE" in med:
            out[stage] = int(round((2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024))
json.dump(out, open("gpurun_out/pmc_epid/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp gpurun_out/pmc_epid/pmc_traffic.json $OUT/pmc_traffic.json
find gpurun_out/pmc_epid -name "*.csv" -size +1M -delete
