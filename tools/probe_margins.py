#!/usr/bin/env python3
"""Margins of the run-level parity tests (convergence frames, converged-SER and noise-estimate differences) for G9 / G10."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_golden
PHI = np.array([0.0314, 0.0314], dtype=np.complex64); TAU = 0.1e-12 * np.sqrt(1000)
from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing as pv
from vae_equalizer_amd.func_VAEflex_DP_MQAM_shaping import processing as pf
conv = lambda s: int(np.argmax((s < 0.1).all(0)))
for name, fn in (("G10_pcs_run", pv), ("G9_flex_run", pf)):
    g = load_golden(name)
    F, N = int(g["num_frames"]), int(g["N_frame_max"])
    nu = float(g["nu"]) if "nu" in g else 0.0
    SER, Var_est, var = fn("64-QAM", 2, 23, nu, 25, float(g["theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0", 90e9, -26e-24, TAU, PHI, 170, seed=int(g["seed"]), verbose=False)
    o, r = SER.numpy(), g["SER"]
    lo = max(conv(o), conv(r)) + 4
    print(name, "conv", conv(o), conv(r), "first2", np.max(np.abs(o[:, :2] - r[:, :2])), "tail diff", np.abs(o[:, lo:].mean(1) - r[:, lo:].mean(1)), "var rel", np.max(np.abs(Var_est.numpy()[:, lo:].mean(1) - g["Var_est"][:, lo:].mean(1)) / g["Var_est"][:, lo:].mean(1)))
