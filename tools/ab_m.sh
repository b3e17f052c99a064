#!/bin/bash
# A/B two library variants over tap counts and minibatch lengths: tools/ab_m.sh "<variants>" "<M list>" "<B list>"
for M in $2; do for B in $3; do for v in $1; do echo -n "M=$M B=$B $v: "; VAEQ_LIB=$PWD/gpurun_variants/libvaeq_$v.so timeout -k 10 200 python tools/probe_scaling.py 0 8192 50 $B $M 2>&1 | grep threads | awk '{print $5, $6, $9, $10, $11}'; done; done; done
