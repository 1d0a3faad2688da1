#!/bin/bash
# SQ counters of the product kernels of one driver command, two rocprofv3 --pmc passes (kernel-trace only beside them).
#   bash scripts/pmc_kernels.sh <tag> <kernel-name-fragment>[,<fragment>...] -- python scripts/run_ct_pass.py 25 2
TAG=$1; FRAGS=$2; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/a -o p -- "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o p -- "$@" > $OUT/b.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: they do not fit one pass); the summary prints
# the raw counter (KiB) and, for FETCH_SIZE, the doubled value the guide prescribes for wide coalesced reads on gfx950
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c -o p -- "$@" > $OUT/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/d -o p -- "$@" > $OUT/d.log 2>&1
python - "$OUT" "$FRAGS" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out, frags = sys.argv[1], sys.argv[2].split(",")
for frag in frags:
    print("==", frag)
    for sub in ("a", "b", "c", "d"):
        acc = collections.defaultdict(list); dur = []
        for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if frag in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if dur: print(f"  pass {sub}: avg duration {sum(dur) / len(dur) / 1e3:.1f} us over {len(dur)} rows")
        for k, v in sorted(acc.items()):
            med = sorted(v)[len(v) // 2]
            extra = f"   = {med * 1024 / 1e6:10.1f} MB raw" + (f", x 2 = {med * 2048 / 1e6:10.1f} MB (gfx950 correction)" if k == "FETCH_SIZE" else "") if k in ("FETCH_SIZE", "WRITE_SIZE") else ""
            print(f"     {k:24s} med={med:16.0f} n={len(v)}{extra}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
