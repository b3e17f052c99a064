#!/bin/bash
# A/B kernel variants over minibatch lengths on the GPU box: tools/ab_shapes.sh "<variant names>" "<B list>"   (8192 runs, 50 steps, probe_scaling.py)
for B in $2; do for v in $1; do echo -n "B=$B $v: "; VAEQ_LIB=$PWD/gpurun_variants/libvaeq_$v.so timeout -k 10 200 python tools/probe_scaling.py 0 8192 50 $B 2>&1 | grep threads | awk '{print $5, $6, $9, $10, $11}'; done; done
