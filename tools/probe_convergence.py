#!/usr/bin/env python3
"""Convergence frame (first frame with all four SERs < 0.1) of the seeded 140-frame DP VAE-LE run of G7_runs for several seeds
and both kernels (wave-per-run, generic): how far rounding-level differences move the escape from the initial plateau."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G7_runs.npz"), allow_pickle=True)
F, N, seed0 = int(g["vaele_num_frames"]), int(g["vaele_N_frame_max"]), int(g["vaele_seed"])
ref = g["vaele_SER"]
print("reference: conv frame", int(np.argmax((ref < 0.1).all(0))), "tail SER", np.round(ref[:, -10:].mean(1), 4))
seeds = [seed0 + k for k in range(8)]
for th in (0, 256):
    runs = [DPRun(23, 0.0, float(g["vaele_theta_diff"]), np.pi / 10, 2.5e-3, 90e9, s) for s in seeds]
    r = run_dp_batch(runs, "64-QAM", 2, 25, 100, N, F, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000),
                     np.array([0.0314, 0.0314], dtype=np.complex64), 170, flex=False, generator="numpy", threads=th)
    S = r["SER"].numpy()
    print("threads", th, "conv frames", [int(np.argmax((S[i] < 0.1).all(0))) for i in range(len(seeds))],
          "tail SER", np.round(S[:, :, -10:].mean((1, 2)), 4), flush=True)
