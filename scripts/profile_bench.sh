#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench (the PMC traffic passes over the full bench hung
# on this pool twice: scripts/pmc_write_pipeline.sh collects them over a minimal driver of the same kernels).
# Outputs under gpurun_out/prof_$1 ; copy the summaries you want judged into profiles/.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity --sustained-steps 0 > $OUT/bench_trace.log 2>&1
find $OUT -type f -exec ls -la {} \;
# keep the per-dispatch traces small: the summaries are what gets committed
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -f $OUT/trace/*kernel_trace.csv
grep '"metric"' $OUT/bench_trace.log | tail -1
