#!/bin/bash
# PMC counters + kernel durations of the Gaussian kernel from the stopwatch binaries given as arguments (64 frames per launch)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for BIN in "$@"; do
OUT=gpurun_out/pmc_g2d_$(basename $BIN); rm -rf $OUT; mkdir -p $OUT
SETS=("GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
      "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
      "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BIN 64 ${G2D_DATA:-1} > $OUT/p$i.log 2>&1
done
echo "== $BIN"
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out=sys.argv[1]
acc=collections.defaultdict(list)
for f in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gauss2d" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
d=[]
for f in glob.glob(out+"/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gauss2d" in row["Kernel_Name"]:
            d.append(float(row["End_Timestamp"])-float(row["Start_Timestamp"]))
if d:
    d.sort(); print(f"kernel duration ns: median {d[len(d)//2]:.0f} min {d[0]:.0f} (n={len(d)}) for 64 frames")
PY
find $OUT -name "*.csv" -size +1M -delete
done
