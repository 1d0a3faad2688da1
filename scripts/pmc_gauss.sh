#!/bin/bash
# PMC view of the Gaussian kernels that serve 16-bit frames (register-window packed-f32 decision path).
# SQ counters only, two passes; every rocprofv3 call is wrapped in `timeout`.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gauss
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/run_g.py <<'PY'
import torch, sys
sys.path.insert(0, ".")
from pylinac_amd import ops
from pylinac_amd.synthetic import epid_open_field_frames
dev = torch.device("cuda:0")
fr = epid_open_field_frames(64, 1024, 1024, device=dev)
for _ in range(3):
    ops.gaussian_filter(fr, 5)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -o p -- python /tmp/run_g.py > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b -o p -- python /tmp/run_g.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, re, collections, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_gauss"
for sub in ("a","b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n=row["Kernel_Name"]
            if "gauss" not in n: continue
            n=re.sub(r"\(anonymous namespace\)::","",n).split("(")[0].replace("void ","")
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for n,d in acc.items():
        print(n)
        for k,v in d.items(): print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
tail -3 $OUT/a.log
find $OUT -name "*.csv" -size +2M -delete
