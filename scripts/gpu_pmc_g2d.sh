#!/bin/bash
# PMC counters of gauss2d_mm from the stopwatch binary (scripts/ubench/g2d_v0*), 64 frames per launch.  $1 = binary, $2 = "mem" for
# the memory set only
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_g2d; rm -rf $OUT; mkdir -p $OUT
BIN=${1:-scripts/ubench/g2d_v0}
SETS=("FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
if [ "$2" != "mem" ]; then
SETS+=("GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
       "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
       "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BIN 64 > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_g2d/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gauss2d" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
tail -2 $OUT/p1.log
find $OUT -name "*.csv" -size +1M -delete
