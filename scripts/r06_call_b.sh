#!/bin/bash
# Round 6, call b: config #5 with the device-side axis fit (no mid-pass host wait) against the previous commit's library
# path is not comparable library-to-library (the change is in the host flow): the pass is timed on this box and set against
# call a's 4.84 ms; plus the phase stopwatch of mask_regions_kernel (-DPL_SR_TIMING build) and the PCIe forms again.
TAG=${1:-r06b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "ctp528 or catphan or dicom or volume or run_from_host" -rf > $OUT/pytest_ct.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_ct.log)" | tee $OUT/summary.txt
for i in 1 2 3; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee $OUT/ct_pass.txt
PYLINAC_HIP_LIB=build/variants/lib_srt.so timeout 300 python scripts/time_sr_phases.py 25 2>&1 | grep -v amdgpu.ids | tee $OUT/sr_phases.txt
timeout 400 bash scripts/profile_configs.sh ctp25 > $OUT/ct_kernel_stats.txt 2>&1
timeout 300 python scripts/time_pcie_inclusive.py 256 10 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie_inclusive.txt
