#!/bin/bash
# Round 6, call f: mask_regions with the row-based plane build (loads back to back, no vector address arithmetic).
TAG=${1:-r06f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "ctp528 or catphan or volume or edge or phantom or regions" -rf > $OUT/pytest_ct.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_ct.log)" | tee $OUT/summary.txt
for i in 1 2 3; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee $OUT/ct_pass.txt
timeout 400 bash scripts/profile_configs.sh ctp25 > $OUT/ct_kernel_stats.txt 2>&1
head -8 $OUT/ct_kernel_stats.txt
