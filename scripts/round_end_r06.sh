#!/bin/bash
# Round 6: one gpurun call that refreshes every piece of evidence the round is judged on; EVERY step under its own time-out
# (r04's final PMC pass sat until gpurun's limit).
#   gpurun --timeout 2400 -- 'bash scripts/round_end_r06.sh r06x'
TAG=${1:-r06x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/summary.txt
# the driver's own invocation
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained"))
print("mfma_view", d["roofline"].get("mfma_view"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("frac"), c.get("parity_sample", {}).get("ok"), c.get("cpu_baseline", {}).get("value"), c.get("cpu_baseline", {}).get("pool"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("pool"))
PY
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run', d['value'], d['ms_per_step'], d['roofline']['stage_ms']['gauss2d'], 'sustained', d['sustained']['ms_per_step'])"; done | tee $OUT/bench_three_runs.txt
# a --gpus 2 request on this one-GPU box: never a one-GPU line
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $OUT/bench_gpus2_stdout.log 2> $OUT/bench_gpus2_stderr.log
echo "bench --gpus 2 on one GPU: rc=$? json lines=$(grep -c '"metric"' $OUT/bench_gpus2_stdout.log)" | tee -a $OUT/summary.txt
timeout 300 bash scripts/profile_bench.sh $TAG > $OUT/profile_bench.log 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary.txt
cp gpurun_out/prof_$TAG/summary.json $OUT/rocprofv3_summary.json 2>/dev/null
grep '"metric"' gpurun_out/prof_$TAG/bench_trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
# HBM traffic of the pipeline kernels: FETCH_SIZE / WRITE_SIZE in separate --pmc passes over a light driver (r05e's method)
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
cat $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
timeout 400 bash scripts/profile_configs.sh > $OUT/configs_kernel_stats.txt 2>&1
# HBM traffic of the Winston-Lutz pass with and without the tile maxima (FETCH / WRITE in separate --pmc passes inside pmc_kernels.sh)
timeout 400 bash scripts/pmc_kernels.sh wl hist16_two_window_kernel,cax_reduce,cax_window,bb_sweep_kernel,cax_reduce_tiles -- python scripts/run_wl_pass.py 512 2 > /dev/null 2>&1
cp gpurun_out/pmc_wl/summary.txt $OUT/pmc_wl_kernels.txt
timeout 400 bash scripts/pmc_kernels.sh pf pf_windows_kernel,scaled_colmeanv,minmax_kernel -- python scripts/run_pf_pass.py 512 2 > /dev/null 2>&1
cp gpurun_out/pmc_pf/summary.txt $OUT/pmc_sq_pf_kernels.txt
{ for mode in "notiles" "" "notiles noise" "noise"; do echo "== run_wl_pass.py 1250 5 $mode"; timeout 300 python scripts/run_wl_pass.py 1250 5 $mode 2>&1 | tail -1; done; } | tee $OUT/wl_pass.txt
{ echo "== exact (float64 np.std for every window)"; timeout 300 python scripts/run_pf_pass.py 512 8 exact | tail -1; echo "== product"; timeout 300 python scripts/run_pf_pass.py 512 8 | tail -1; } | tee $OUT/pf_pass.txt
timeout 300 python scripts/run_ct_pass.py 25 8 | tee $OUT/ct_pass.txt
timeout 300 python scripts/time_pcie_inclusive.py 256 10 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie_inclusive.txt
for f in 256 64 32 8; do timeout 120 python bench.py --gpus 1 --frames $f --steps 30 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames', $f, 'ms/step', d['ms_per_step'], 'us/frame', round(d['ms_per_step']*1e3/$f, 3), d['roofline']['stage_ms'])"; done | tee $OUT/small_batch_sweep.txt
# round 6 additions: config #5's SQ counters + HBM traffic on the final build
timeout 500 bash scripts/pmc_kernels.sh ct mask_regions_kernel,edge_otsu_kernel,edge_stream32_kernel,circle_profile_combined,circle_ring_kernel -- python scripts/run_ct_pass.py 25 2 > /dev/null 2>&1
cp gpurun_out/pmc_ct/summary.txt $OUT/pmc_sq_ct_kernels.txt
# gpurun copies back at most 64 MiB: the raw profiler trees are scratch once their summaries sit in $OUT (r06zw: a timed-out PMC
# pass left its traces behind and NOTHING came back)
for d in gpurun_out/*/; do [ "$d" = "gpurun_out/$TAG/" ] || rm -rf "$d"; done
du -sh gpurun_out | tee -a $OUT/summary.txt
