#!/bin/bash
# instruction-cache / co-execution counters of gauss2d_mm (stopwatch binary, 256 frames, few launches) and of the MFMA / VALU microbenchmark
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_g2d3; rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"
      "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/g$i -o p -- scripts/ubench/g2d_v0 256 1 4 > $OUT/g$i.log 2>&1
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/u$i -o p -- scripts/ubench/mfma_valu_settle 4 1 > $OUT/u$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, re
out=sys.argv[1]
for tag in ("g","u"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out+f"/{tag}*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n=re.sub(r"\(anonymous namespace\)::","",row["Kernel_Name"]).split("(")[0].replace("void ","")
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for n,d in acc.items():
        print(n)
        for k,v in sorted(d.items()): print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find $OUT -name "*.csv" -size +1M -delete
