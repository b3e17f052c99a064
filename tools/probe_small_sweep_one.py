#!/usr/bin/env python3
"""One small sweep through run_dp_batch for a kernel trace: tools/probe_small_sweep_one.py [runs] [frames]  (VAEQ_SERIAL_FRAMES=1 for the serial order)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
F = int(sys.argv[2]) if len(sys.argv) > 2 else 60
NU, SNR = [0, 0.0270955, 0.0872449, 0.1222578], [20, 22, 24, 26, 28]
runs = [DPRun(s, nu, 0.06 * np.pi, np.pi / 10, lr, 90e9) for nu in NU for lr in (2.5e-3, 2e-3, 3e-3) for s in SNR for i in range(5)][:R]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_dp_batch(runs, "64-QAM", 2, 25, 100, 10000, F, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64), 170)
    torch.cuda.synchronize()
    print(f"{len(runs)} runs x {F} frames: {1e3 * (time.perf_counter() - t0) / F:.3f} ms per frame", flush=True)
