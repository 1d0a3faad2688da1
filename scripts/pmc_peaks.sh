#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_peaks
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/a -o p -- python scripts/time_peaks.py > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b -o p -- python scripts/time_peaks.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_peaks"
for sub in ("a","b"):
    acc=collections.defaultdict(list); dur=[]
    for f in glob.glob(f"{out}/{sub}/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "find_peaks_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur.append(int(row["End_Timestamp"])-int(row["Start_Timestamp"]))
    print(sub, "avg dur ns", sum(dur)/max(len(dur),1))
    for k,v in acc.items(): print(f"   {k:24s} min={min(v):14.0f} med={sorted(v)[len(v)//2]:14.0f} max={max(v):14.0f} n={len(v)}")
PY
