#!/bin/bash
# fixed-layout (default) vs run-time-layout (VAEQ_DP_RUNTIME_LAYOUT=1) instantiations of the DP wave kernel over minibatch lengths, same box:
# tools/ab_layout.sh "<B list>"   (8192 runs, 50 steps, probe_scaling.py)
for B in $1; do for v in 0 1; do echo -n "B=$B runtime_layout=$v: "; VAEQ_DP_RUNTIME_LAYOUT=$v timeout -k 10 200 python tools/probe_scaling.py 0 8192 50 $B 2>&1 | grep threads | awk '{print $5, $6, $9, $10, $11}'; done; done
