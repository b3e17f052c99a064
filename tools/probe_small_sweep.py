#!/usr/bin/env python3
"""Per-frame wall time of small sweeps through run_dp_batch (device generator): the reference's DEFAULT sweep (15 runs = 3 learning rates x 5
seeds) and config 5's script-faithful grid size (300 runs), frames on three streams (default below the resident-run count) vs serial
(VAEQ_SERIAL_FRAMES=1).   tools/probe_small_sweep.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NU, SNR = [0, 0.0270955, 0.0872449, 0.1222578], [20, 22, 24, 26, 28]
sets = {15: [DPRun(23, 0.0, 0.06 * np.pi, np.pi / 10, lr, 90e9) for lr in (2.5e-3, 2e-3, 3e-3) for i in range(5)],
        300: [DPRun(s, nu, 0.06 * np.pi, np.pi / 10, lr, 90e9) for nu in NU for lr in (2.5e-3, 2e-3, 3e-3) for s in SNR for i in range(5)],
        38: [DPRun(s, nu, 0.06 * np.pi, np.pi / 10, lr, 90e9) for nu in NU for lr in (2.5e-3, 2e-3, 3e-3) for s in SNR for i in range(5)][::8]}
for R, runs in sets.items():
    for mode in ("overlap", "serial", "overlap", "serial"):
        os.environ.pop("VAEQ_SERIAL_FRAMES", None)
        if mode == "serial":
            os.environ["VAEQ_SERIAL_FRAMES"] = "1"
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 10000, F, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64), 170)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"{len(runs):4d} runs x {F} frames x 10000 symbols, {mode:8s}: {1e3 * (t1 - t0) / F:.3f} ms per frame", flush=True)
