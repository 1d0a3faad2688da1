#!/bin/bash
# Round 5, second GPU call: the -m gpu suite on the new kernels, then A/B on one box: picket-fence window kernel (r05 start /
# integer-moment edge test / + one-wave FWXM short cut), the Winston-Lutz pass with the one-launch record table, the
# PCIe-inclusive headline step, and the bench line.
TAG=${1:-r05b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
# picket-fence window kernel, same box: (1) numpy's float64 deviations for every window (= the kernel r05 started with),
# (2) the integer-moment edge test without the one-wave FWXM short cut (build/variants/lib_nofast.so), (3) the product build
pfrun() {  # $1 = label, $2.. = extra args of run_pf_pass.py
  echo "== $1"; shift
  python scripts/run_pf_pass.py 256 20 "$@" | tail -1
  python scripts/run_pf_pass.py 256 20 "$@" | tail -1
  rm -rf /tmp/prof_ab; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python scripts/run_pf_pass.py 256 4 "$@" > /dev/null 2>&1
  python - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/prof_ab/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(r"colmean|minmax_kernel<unsigned short>|pf_windows", r["Name"]):
            print(f'   {r["Name"][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
}
cp pylinac_amd/libpylinac_hip.so /tmp/lib_prod.so
{ pfrun "product library, exact_deviation=True (float64 np.std sequence for every window)" exact
  cp build/variants/lib_nofast.so pylinac_amd/libpylinac_hip.so
  pfrun "integer-moment edge test, FWXM search without the one-wave short cut (-DPL_FWXM_FAST=0)"
  cp /tmp/lib_prod.so pylinac_amd/libpylinac_hip.so
  pfrun "product library"; } 2>&1 | grep -v amdgpu.ids | tee $OUT/pf_window_variants.txt
# the WL driver needs the host code of the matching tree: old library with the old torch.stack path is what r05a's bench line has
# (617 k frames/s); here only the new tree is timed, twice
for i in 1 2; do timeout 300 python scripts/run_wl_pass.py 1250 5 2>&1 | tail -1; done | tee $OUT/wl_pass.txt
timeout 300 python scripts/time_pcie_inclusive.py 256 10 2>&1 | tail -1 | tee $OUT/pcie_inclusive.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained", {}).get("value"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("parity_sample", {}).get("ok"))
PY
