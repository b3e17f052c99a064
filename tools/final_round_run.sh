cd /root/repo
rm -f gpurun_out/r03_ensemble_parity.txt; VAEQ_ENSEMBLE_LOG=gpurun_out/r03_ensemble_parity.txt python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests.log 2>&1; echo rc=$? >> gpurun_out/r03_gputests.log; tail -4 gpurun_out/r03_gputests.log
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err; tail -c 600 gpurun_out/r03_bench_line.json
python bench.py --config5 --no-cpu-baseline > gpurun_out/r03_bench_config5_line.json 2>> gpurun_out/r03_bench.err; tail -c 300 gpurun_out/r03_bench_config5_line.json
python bench.py --config5 --iter 5 --no-cpu-baseline > gpurun_out/r03_bench_config5_iter5_line.json 2>> gpurun_out/r03_bench.err
tools/profile_round.sh r03prof > gpurun_out/r03_profile_round.log 2>&1; tail -8 gpurun_out/r03_profile_round.log
tools/profile_pmc.sh dp 8192 /root/repo/gpurun_out/r03prof/pmc_dp > gpurun_out/r03_pmc_dp.log 2>&1; tail -6 gpurun_out/r03_pmc_dp.log
python tools/probe_small_sweep.py 60 > gpurun_out/r03_small_sweep.txt 2>&1; cat gpurun_out/r03_small_sweep.txt
python tools/probe_pipeline.py 8192 compact > gpurun_out/r03_pipeline_probe.txt 2>&1; cat gpurun_out/r03_pipeline_probe.txt
tools/profile_pmc.sh awgn 8192 /root/repo/gpurun_out/r03prof/pmc_awgn > gpurun_out/r03_pmc_awgn.log 2>&1; tail -3 gpurun_out/r03_pmc_awgn.log
tools/profile_pmc_nn.sh 2048 /root/repo/gpurun_out/r03prof/pmc_nn > gpurun_out/r03_pmc_nn.log 2>&1; tail -4 gpurun_out/r03_pmc_nn.log
