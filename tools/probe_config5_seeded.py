#!/usr/bin/env python3
"""The reduced-size on-grid config-5 points on the frames the reference saw (G13_cfg5_nu*): ours vs reference per-frame statistics.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing
PHI = np.array([0.0314, 0.0314], dtype=np.complex64)
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for f in ("G13_cfg5_nu0872_snr20", "G13_cfg5_nu1222_snr28"):
    g = np.load(os.path.join(gd, f + ".npz"))
    F, N = int(g["num_frames"]), int(g["N_frame_max"])
    SER, Var_est, var = processing("64-QAM", 2, float(g["SNR"]), float(g["nu"]), 25, float(g["theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0", 90e9,
                                   -26e-24, 0.1e-12 * np.sqrt(1000), PHI, 170, seed=int(g["seed"]), verbose=False)
    o, r = SER.numpy(), g["SER"]
    print(f, "var", var.numpy(), g["var"])
    for a, b in ((0, 20), (20, 60), (60, 120), (120, 200)):
        print(f"  frames {a:3d}-{b:3d}: SER ours {np.round(o[:, a:b].mean(1), 3)} ref {np.round(r[:, a:b].mean(1), 3)}  Var_est ours {np.round(Var_est.numpy()[:, a:b].mean(1), 5)} ref {np.round(g['Var_est'][:, a:b].mean(1), 5)}")
    print("  max |dSER| first 5 frames", np.abs(o[:, :5] - r[:, :5]).max(), " per-frame |dVar|/Var first 5", np.abs(Var_est.numpy()[:, :5] - g["Var_est"][:, :5]).max() / g["Var_est"][:, :5].max())
