set -e
timeout -k 10 600 python -m pytest tests/test_awgn_kernel_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
  for v in awgn_cf0 SHIPPED; do
    if [ $v = SHIPPED ]; then unset VAEQ_LIB; else export VAEQ_LIB=$PWD/gpurun_variants/libvaeq_$v.so; fi
    echo "== $v (rep $rep)"; timeout -k 10 200 python tools/probe_awgn.py 8192 30 0 2>&1 | grep threads
  done
done
