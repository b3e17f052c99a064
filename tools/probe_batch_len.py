#!/usr/bin/env python3
"""Drop-in DP pipeline (on-device generator -> training kernel -> epilogue) at several minibatch lengths: the multi-wave kernels
(B > 128) must converge like B = 100.  32 runs x 170 frames of 10 000 symbols, 64-QAM, SNR 23 dB, h0, static polarisation state unless argv[2] gives the drift per frame in units of pi.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch

R, frames = 32, 170
td = float(sys.argv[2]) * np.pi if len(sys.argv) > 2 else 0.0     # per-frame polarisation drift (the sweep script's default: 0.06 pi)
for B in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["100", "200", "400", "1000"])]:
    runs = [DPRun(23.0, 0.0, td, np.pi / 10, 2.5e-3, 90e9, seed=100 + i) for i in range(R)]
    t0 = time.time()
    r = run_dp_batch(runs, "64-QAM", 2, 25, B, 10000, frames, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64),
                     170, generator="hip", device="cuda:0")
    torch.cuda.synchronize()
    ser = r["SER"].numpy()[:, :2]                                # [R, pol, frame]
    tail = ser[:, :, -10:].mean(-1).max(-1)
    print(f"B={B:5d}: {time.time() - t0:6.2f} s  locked runs (tail SER < 0.05): {(tail < 0.05).sum():3d}/{R}  median tail SER {np.median(tail):.4f}", flush=True)
