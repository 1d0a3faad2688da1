#!/bin/bash
# power-cap probe: the same kernel on EPID-like, full-range and low-toggle data, sustained, with rocm-smi power / sclk samples
cd $GRAFT_REPO_ROOT
for d in 1 0 2; do
  scripts/ubench/g2d_v0 256 $d 5000 > /tmp/g2d.out &
  PID=$!
  sleep 2.0
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/=*//g; s/GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 0.2; done
  wait $PID; cat /tmp/g2d.out
done
