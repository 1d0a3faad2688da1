#!/bin/bash
# Round 4: one gpurun call that refreshes every piece of evidence the round is judged on.
#   gpurun --timeout 1800 -- 'bash scripts/round_end_r04.sh r04z'
# Every step carries its own time-out: the r04y run sat in a --pmc pass until gpurun's limit and brought nothing back.
TAG=${1:-r04z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=5 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/summary.txt
# the driver's own invocation
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained"))
for k, c in d.get("configs", {}).items():
    print(k, c["value"], c["unit"], c["ms_per_pass"], c["parity_sample"]["ok"], c.get("cpu_baseline", {}).get("value"), c.get("cpu_baseline", {}).get("pool"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("pool"))
PY
# three more bench lines on the same box (box-to-box spread is +-10 %, run-to-run on one box < 1 %)
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run', d['value'], d['ms_per_step'], d['roofline']['stage_ms']['gauss2d'], 'sustained', d['sustained']['ms_per_step'])"; done | tee $OUT/bench_three_runs.txt
timeout 300 bash scripts/profile_bench.sh $TAG > $OUT/profile_bench.log 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary.txt
cp gpurun_out/prof_$TAG/summary.json $OUT/rocprofv3_summary.json 2>/dev/null
grep '"metric"' gpurun_out/prof_$TAG/bench_trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
timeout 300 bash scripts/pmc_write_pipeline.sh > $OUT/pmc_pipeline_traffic.txt 2>&1
cp gpurun_out/pmc_pipe/pmc_traffic.json $OUT/pmc_traffic.json
cat $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
timeout 300 bash scripts/gpu_pmc_bench.sh > $OUT/pmc_sq_pipeline_kernels.txt 2>&1
timeout 400 bash scripts/profile_configs.sh > $OUT/configs_kernel_stats.txt 2>&1
# SQ counters of the config #5 / #3 kernels (two --pmc passes each, kernel trace only beside them)
timeout 300 bash scripts/pmc_kernels.sh ct edge_otsu_kernel,edge_stream_kernel,mask_regions_kernel,circle_profile_combined,peak_valley_kernel -- python scripts/run_ct_pass.py 25 2 > /dev/null 2>&1
cp gpurun_out/pmc_ct/summary.txt $OUT/pmc_sq_ct_kernels.txt
timeout 300 bash scripts/pmc_kernels.sh pf pf_windows_kernel,scaled_colmeanv,minmax_kernel -- python scripts/run_pf_pass.py 512 2 > /dev/null 2>&1
cp gpurun_out/pmc_pf/summary.txt $OUT/pmc_sq_pf_kernels.txt
python scripts/run_ct_pass.py 25 8 | tee -a $OUT/summary.txt
python scripts/run_pf_pass.py 512 8 | tee -a $OUT/summary.txt
python scripts/time_wide_range.py 2>&1 | tail -2 | tee $OUT/wide_range.txt
python scripts/time_hill_batch.py 4096 200 2>&1 | grep -v amdgpu.ids | tee $OUT/hill_batch.txt
python scripts/time_elementwise.py 2>&1 | grep -v amdgpu.ids | tee $OUT/elementwise.txt
for f in 256 64 32 8; do timeout 120 python bench.py --gpus 1 --frames $f --steps 30 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames', $f, 'ms/step', d['ms_per_step'], 'us/frame', round(d['ms_per_step']*1e3/$f, 3), d['roofline']['stage_ms'])"; done | tee $OUT/small_batch_sweep.txt
