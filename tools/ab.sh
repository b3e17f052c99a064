#!/bin/bash
# A/B the kernel variants under gpurun_variants/ on the GPU box: tools/ab.sh "<names>" "<run counts>"
for rep in 1 2; do for v in $1; do echo "== $v (rep $rep)"; VAEQ_LIB=$PWD/gpurun_variants/libvaeq_$v.so timeout -k 10 200 python tools/probe_scaling.py 1 $2 100 2>&1 | grep threads; done; done
