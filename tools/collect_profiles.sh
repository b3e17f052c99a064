#!/bin/bash
# Copy what tools/final_round_run.sh left under gpurun_out/ (scratch) into profiles/<round>/ (tracked): tools/collect_profiles.sh r03
# Run in the build container after the gpurun call has merged its gpurun_out/ back.
set -e
cd $(dirname $0)/..
r=${1:-r03}; G=gpurun_out; P=profiles/$r; O=$G/${r}prof
mkdir -p $P
tail -3 $G/${r}_gputests.log | grep -v amdgpu.ids > $P/gpu_tests_summary.txt
cp $G/${r}_bench_line.json $P/bench_line.json
cp $G/${r}_bench_config5_line.json $P/bench_config5_line.json
cp $G/${r}_bench_config5_iter5_line.json $P/bench_config5_iter5_line.json
cp $G/${r}_ensemble_parity.txt $P/ensemble_parity.txt
cp $O/bench/bench_kernel_stats.csv $P/bench_kernel_stats.csv
grep '^{"metric"' $O/bench.log | tail -1 > $P/bench_profiled_line.json
cp $O/dp_pipe/dp_pipe_kernel_stats.csv $P/dp_pipeline_kernel_stats.csv
cp $O/awgn/awgn_kernel_stats.csv $P/awgn_pipeline_kernel_stats.csv
cp $O/nn/nn_kernel_stats.csv $P/vaenn_kernel_stats.csv
cp $O/pmc_traffic.json $P/pmc_traffic.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cp $O/traffic/pmc_fetch_dp_wave_kernel.csv $O/traffic/pmc_write_dp_wave_kernel.csv $P/
cp $O/pmc_dp/summary.txt $P/dp_wave_pmc_summary.txt
cp $O/pmc_awgn/summary.txt $P/awgn_wave_pmc_summary.txt
{ grep '^#' $P/vaenn_pmc_summary.txt 2>/dev/null || echo "# tools/profile_pmc_nn.sh 2048 (SQ counters of nn_train_kernel, means over the launches)"; grep -v amdgpu.ids $G/${r}_pmc_nn.log | grep 'n=' ; } > $P/vaenn_pmc_summary.txt.new && mv $P/vaenn_pmc_summary.txt.new $P/vaenn_pmc_summary.txt
{
  echo "# tools/probe_pipeline.py 8192 compact (stage by stage, a synchronisation after each)"
  grep "^R=" $G/${r}_pipeline_probe.txt
  echo "# tools/probe_awgn_pipeline.py 8192 (under rocprofv3 --kernel-trace in tools/profile_round.sh)"
  grep "^R=" $O/awgn.log
  echo "# tools/probe_nn.py 2048 (under rocprofv3 --kernel-trace in tools/profile_round.sh)"
  grep "^R=" $O/nn.log
  echo "# tools/probe_small_sweep.py 60 (run_dp_batch, device generator; frames on three streams vs serial)"
  grep "runs x" $G/${r}_small_sweep.txt
} > $P/pipeline_probes.txt
git status --short $P profiles/pmc_traffic.json
