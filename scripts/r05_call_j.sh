#!/bin/bash
# Round 5, call j: does the BB sweep gain from a third workgroup per CU?  the previous commit (build/variants/lib_prev.so: two-tier sweep, 79 KB of LDS)
# (-DPL_SW_FIRST_RUNS=640 -DPL_SW_MAX_CROP=48: 50 KB of LDS instead of 79) against the product, one box.
TAG=${1:-r05j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
stats() {
  python - "$1" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:7]:
    if "at::native" not in r["Name"]:
        print(f'   {r["Name"][:80]:80s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f} min={float(r["MinNs"])/1e3:8.1f} max={float(r["MaxNs"])/1e3:8.1f}')
PY
}
{ for lib in "" build/variants/lib_prev.so "" build/variants/lib_prev.so; do
    echo "== run_wl_pass.py 1250 5, library: ${lib:-product}"
    for i in 1 2; do PYLINAC_HIP_LIB=${lib:-$GRAFT_REPO_ROOT/pylinac_amd/libpylinac_hip.so} timeout 300 python scripts/run_wl_pass.py 1250 5 | tail -1; done
    echo "== noise"
    PYLINAC_HIP_LIB=${lib:-$GRAFT_REPO_ROOT/pylinac_amd/libpylinac_hip.so} timeout 300 python scripts/run_wl_pass.py 1250 5 noise | tail -1
  done
  for frames in 512 1250; do for lib in "" build/variants/lib_prev.so; do
    echo "== kernel stats, $frames frames, library: ${lib:-product}"
    rm -rf /tmp/prof_wl; PYLINAC_HIP_LIB=${lib:-$GRAFT_REPO_ROOT/pylinac_amd/libpylinac_hip.so} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o p -- python scripts/run_wl_pass.py $frames 4 > /dev/null 2>&1
    stats /tmp/prof_wl
  done; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/wl_sweep_occupancy_ab.txt
