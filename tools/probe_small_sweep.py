#!/usr/bin/env python3
"""Wall time of the reference's DEFAULT DP sweep size (15 runs: 3 learning rates x 5 seeds) for a few frames: launch-bound regime."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
runs = [DPRun(23, 0.0, 0.06 * np.pi, np.pi / 10, lr, 90e9, 100 + i) for lr in (2.5e-3, 2e-3, 3e-3) for i in range(5)]
for gen in ("hip", "numpy"):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 10000, 10, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64),
                         170, generator=gen)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"generator={gen}: 15 runs x 10 frames x 10000 symbols: {1e3 * (t1 - t0) / 10:.2f} ms per frame", flush=True)
