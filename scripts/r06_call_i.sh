#!/bin/bash
# Round 6, call i: the packed-float32 edge kernel (pl_edge_plane32) -- the GPU suite's CatPhan tests, then the config #5 pass and
# its kernel statistics with the knob on (product) and off (PL_EDGE_PLANE32=0: the exact float64 edge_stream kernel).
TAG=${1:-r06i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s -k "ctp528 or catphan or volume or edge or phantom or regions or circle or bench_size" -rf 2>&1 | tail -6 | tee $OUT/summary.txt
for knob in 1 0; do
  export PL_EDGE_PLANE32=$knob
  echo "== PL_EDGE_PLANE32=$knob" | tee -a $OUT/summary.txt
  for i in 1 2 3; do timeout 300 python scripts/run_ct_pass.py 25 8; done | tee -a $OUT/summary.txt
  timeout 400 bash scripts/profile_configs.sh ctp25 2>&1 | head -9 | tee -a $OUT/summary.txt
done
