#!/bin/bash
# Round 6, call l: A/B on one box -- the product library (circle profile: the ring's largest radius looked up by the eight radius lanes
# together) against the previous commit's build (lib_circold2.so: twenty loads per lane in front of the taps).
TAG=${1:-r06l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
for lib in "" build/variants/lib_circold2.so; do
  export PYLINAC_HIP_LIB=$lib; [ -z "$lib" ] && unset PYLINAC_HIP_LIB
  echo "== library: ${lib:-product}" | tee -a $OUT/summary.txt
  timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "ctp528 or catphan or volume or edge or phantom or regions or circle" -rf 2>&1 | tail -1 | tee -a $OUT/summary.txt
  for i in 1 2 3; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | head -8 | tee -a $OUT/summary.txt
done
