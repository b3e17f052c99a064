#!/bin/bash
# Per-round profile collection on the GPU box: tools/profile_round.sh <outdir under gpurun_out>
# 1. bench default under --kernel-trace --stats   2. AWGN sweep pipeline (config-2 shape) under --kernel-trace --stats
# 3. VAE-NN kernels (tools/probe_nn.py) under --kernel-trace --stats
# 4. PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, no other trace domain) for the AWGN training kernel
OUT=/root/repo/gpurun_out/${1:-prof_round}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 /root/repo/bench.py --steps 20 --warmup 2 --min-seconds 0 --no-parity --no-extras --no-cpu-baseline > $OUT/bench.log 2>&1 || { echo bench failed; tail -5 $OUT/bench.log; exit 1; }
tail -1 $OUT/bench.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/awgn -o awgn -- python3 /root/repo/tools/probe_awgn_pipeline.py 8192 > $OUT/awgn.log 2>&1 || { echo awgn failed; tail -5 $OUT/awgn.log; exit 1; }
grep "R=" $OUT/awgn.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dp_pipe -o dp_pipe -- python3 /root/repo/tools/probe_pipeline.py 8192 compact > $OUT/dp_pipe.log 2>&1 || { echo dp_pipe failed; tail -5 $OUT/dp_pipe.log; exit 1; }
grep "R=" $OUT/dp_pipe.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nn -o nn -- python3 /root/repo/tools/probe_nn.py 2048 > $OUT/nn.log 2>&1 || { echo nn failed; tail -5 $OUT/nn.log; exit 1; }
grep "R=" $OUT/nn.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 /root/repo/tools/probe_awgn.py 8192 30 0 > $OUT/pmc_$c.log 2>&1 || { echo pmc $c failed; tail -5 $OUT/pmc_$c.log; exit 1; }
done
# HBM traffic of the bench kernel from the PMC counters -> profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic): regenerated with every
# round's final build so that it cannot go stale (kernel name and run count are checked by bench.py)
/root/repo/tools/profile_traffic.sh $(basename $OUT)/traffic && cp $OUT/traffic/pmc_traffic.json $OUT/pmc_traffic.json
echo done
