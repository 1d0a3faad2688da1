#!/bin/bash
# PMC comparison of the Gaussian kernels (fused vs unfused pipeline), SQ counters only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gauss
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/run_both.py <<'PY'
import torch, sys
sys.path.insert(0, ".")
from pylinac_amd.pipeline import EpidPipeline
from pylinac_amd.synthetic import epid_open_field_frames
dev = torch.device("cuda:0")
fr = epid_open_field_frames(64, 1024, 1024, device=dev)
for fused in (True, False):
    p = EpidPipeline(64, 1024, 1024, dev, fused=fused)
    for _ in range(2):
        p.run(fr)
    torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -o p -- python /tmp/run_both.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b -o p -- python /tmp/run_both.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, re, collections, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_gauss"
for sub in ("a","b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            n=row["Kernel_Name"]
            if "gauss" not in n and "median3" not in n: continue
            n=re.sub(r"\(anonymous namespace\)::","",n).split("(")[0].replace("void ","")
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for n,d in acc.items():
        print(n)
        for k,v in d.items(): print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
tail -3 $OUT/a.log
