#!/bin/bash
# SQ / LDS counters of every product kernel of the EPID pipeline (two rocprofv3 --pmc passes over a short bench run)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bench; rm -rf $OUT; mkdir -p $OUT
SETS=("GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python bench.py --no-cpu-baseline --no-configs --no-parity --steps 4 --warmup 2 --sustained-steps 0 > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, re
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n=row["Kernel_Name"]
        if not any(k in n for k in ("gauss2d", "otsu16_window", "median3_threshold", "find_peaks")): continue
        n=re.sub(r"\(anonymous namespace\)::","",n).split("(")[0].replace("void ","")
        acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur=collections.defaultdict(list)
for f in glob.glob(out+"/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n=re.sub(r"\(anonymous namespace\)::","",row["Kernel_Name"]).split("(")[0].replace("void ","")
        if n in acc: dur[n].append(float(row["End_Timestamp"])-float(row["Start_Timestamp"]))
for n,d in acc.items():
    dd=sorted(dur[n]); print(n, "duration ns median", dd[len(dd)//2] if dd else None)
    for k,v in sorted(d.items()): print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find $OUT -name "*.csv" -size +1M -delete
