#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-c2}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf 2>&1 | tail -8 | tee $OUT/pytest.txt
bash scripts/profile_configs.sh wln wl > $OUT/configs_kernel_stats.txt 2>&1
grep -v "at::native" $OUT/configs_kernel_stats.txt | cut -c1-200 | head -30
