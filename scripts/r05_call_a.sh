#!/bin/bash
# Round 5, first GPU call: the new code on the device (full -m gpu suite, smoke, the driver's bench line), then the stopwatch
# A/B of pf_windows_kernel's stages (build/variants/lib_pfv{1,2,3}.so: no deviation stage / no column median / no FWXM search).
TAG=${1:-r05a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# a box whose first torch kernel faults (r05a's first attempt: "Memory access fault by GPU node-2 ... address 0x3000" inside
# torch.rand, before any kernel of this repository ran) is not worth another second
timeout 300 python -c "import torch; x = torch.rand(1 << 20, device='cuda'); print('torch sanity', float(x.sum()))" || { echo "BAD BOX: torch's own kernel faults"; exit 7; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --durations=8 > $OUT/pytest_gpu_full.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu_full.log)" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.log 2>$OUT/bench_stderr.log ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/bench_line_full.json
python - $OUT/bench_line_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"], d.get("parity_sample", {}).get("ok"), "sustained", d.get("sustained", {}).get("value"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("unit"), c.get("ms_per_pass"), c.get("parity_sample", {}).get("ok"))
PY
# plain `python bench.py --gpus 2` on a one-GPU box: must exit non-zero (rank 1 has no device), never print a 1-GPU line
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $OUT/bench_gpus2_stdout.log 2> $OUT/bench_gpus2_stderr.log
echo "bench --gpus 2 on one GPU: rc=$? lines=$(grep -c '"metric"' $OUT/bench_gpus2_stdout.log)" | tee -a $OUT/summary.txt
tail -5 $OUT/bench_gpus2_stderr.log >> $OUT/summary.txt
timeout 600 bash scripts/gpu_ab_pf_kernels.sh pylinac_amd/libpylinac_hip.so build/variants/lib_pfv1.so build/variants/lib_pfv2.so build/variants/lib_pfv3.so 2>&1 | grep -v amdgpu.ids | tee $OUT/pf_stage_stopwatch.txt
