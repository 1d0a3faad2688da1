#!/bin/bash
# the bench line three times on one box (run-to-run spread), stage times only
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-configs --steps 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"
done 2>&1 | tee gpurun_out/bench3.txt
