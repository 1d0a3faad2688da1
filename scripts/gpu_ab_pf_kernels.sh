#!/bin/bash
# Development aid: A/B of library builds on the config #3 pass, with rocprofv3 averages of its streaming kernels:
#   bash scripts/gpu_ab_pf_kernels.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
cp pylinac_amd/libpylinac_hip.so /tmp/lib_orig.so
export TMPDIR=/tmp
for lib in "$@"; do
  cp $lib pylinac_amd/libpylinac_hip.so
  echo "== $(basename $lib)"
  python scripts/run_pf_pass.py 256 20 | tail -1
  python scripts/run_pf_pass.py 256 20 | tail -1
  rm -rf /tmp/prof_ab; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python scripts/run_pf_pass.py 256 4 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/prof_ab/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(r"colmean|minmax_kernel<unsigned short>|pf_windows", r["Name"]):
            print(f'   {r["Name"][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
done
cp /tmp/lib_orig.so pylinac_amd/libpylinac_hip.so
