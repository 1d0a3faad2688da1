#!/bin/bash
# Round 6, call e: mask_regions' phase stopwatch on the bit-domain build; SQ counters of config #5's kernels.
TAG=${1:-r06e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
PYLINAC_HIP_LIB=build/variants/lib_srt.so timeout 300 python scripts/time_sr_phases.py 25 2>&1 | grep -v amdgpu.ids | tee $OUT/sr_phases.txt
timeout 500 bash scripts/pmc_kernels.sh ct mask_regions_kernel,edge_otsu_kernel,edge_stream_kernel,circle_profile_combined -- python scripts/run_ct_pass.py 25 2 > /dev/null 2>&1
cp gpurun_out/pmc_ct/summary.txt $OUT/pmc_sq_ct_kernels.txt
cat $OUT/pmc_sq_ct_kernels.txt | head -80
