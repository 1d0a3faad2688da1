#!/bin/bash
# what do the shader clock and the socket power do under gauss2d_mm8?  polls rocm-smi while the stopwatch binary loops for ~3 s
cd $GRAFT_REPO_ROOT
scripts/ubench/g2d_v0_w8 256 1 6000 > /tmp/g2d.out &
PID=$!
sleep 2.2
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/=*//g' | tr '\n' ' '; echo; sleep 0.25; done
wait $PID; cat /tmp/g2d.out
