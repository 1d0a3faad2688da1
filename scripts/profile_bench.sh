#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two PMC passes of the default bench.
# Outputs under gpurun_out/prof_$1 ; copy the summaries you want judged into profiles/.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/bench_trace.log 2>&1
timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $OUT/bench_fetch.log 2>&1
timeout 100 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $OUT/bench_write.log 2>&1
find $OUT -type f -exec ls -la {} \;
# keep the per-dispatch traces small: the summaries are what gets committed
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -f $OUT/trace/*kernel_trace.csv
grep '"metric"' $OUT/bench_trace.log | tail -1
