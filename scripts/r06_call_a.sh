#!/bin/bash
# Round 6, call a: the GPU suite on the round's first build (DICOM decode, run_from_host, ADVICE fixes), the PCIe-inclusive
# forms, and this box's baseline for config #5 (per-kernel stats) before the round's kernel work.
TAG=${1:-r06a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=8 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
timeout 300 python scripts/time_pcie_inclusive.py 256 10 2>&1 | grep -v amdgpu.ids | tee $OUT/pcie_inclusive.txt
timeout 300 python scripts/run_ct_pass.py 25 8 | tee $OUT/ct_pass.txt
timeout 400 bash scripts/profile_configs.sh ctp25 > $OUT/ct_kernel_stats.txt 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $OUT/bench_line.json
