#!/usr/bin/env python3
"""Where the host spends its time per frame of a small sweep (run_dp_batch, 300 runs, three streams): cProfile of the enqueueing thread.
tools/probe_host_overhead.py [frames]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NU, SNR = [0, 0.0270955, 0.0872449, 0.1222578], [20, 22, 24, 26, 28]
runs = [DPRun(s, nu, 0.06 * np.pi, np.pi / 10, lr, 90e9) for nu in NU for lr in (2.5e-3, 2e-3, 3e-3) for s in SNR for i in range(5)]
args = (runs, "64-QAM", 2, 25, 100, 10000, F, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64), 170)
run_dp_batch(*args); torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
r = run_dp_batch(*args)
t1 = time.perf_counter()
pr.disable()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{len(runs)} runs x {F} frames: host returned after {1e3 * (t1 - t0) / F:.3f} ms per frame (profiled), device done after {1e3 * (t2 - t0) / F:.3f} ms per frame")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
