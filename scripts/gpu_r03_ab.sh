#!/bin/bash
# A/B of library variants (build/variants/lib_*.so) + the GPU parity tests that touch the changed kernels
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "${KEXPR:-median or otsu or pipeline or epid or headline or fused or bench_line}" 2>&1 | tail -5 | tee $OUT/pytest.txt
STEPS=40 WARM=20 bash scripts/gpu_ab_libs.sh build/variants/lib_*.so 2>&1 | tee $OUT/ab.txt
