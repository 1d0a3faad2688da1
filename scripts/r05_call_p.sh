#!/bin/bash
# Round 5, call p: config #3 staggered over two streams (scripts/run_pf_pass_overlap.py), 2 / 4 / 8 chunks, 512 and 256 frames.
TAG=${1:-r05p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
for n in 512 256; do for c in 2 4 8; do echo "== $n frames, $c chunks"; timeout 200 python scripts/run_pf_pass_overlap.py $n 20 $c 2>&1 | grep -v amdgpu.ids; done; done | tee $OUT/pf_overlap.txt
