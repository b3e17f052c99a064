#!/usr/bin/env python3
"""Run the seeded DP VAE-LE processing() of tests/test_processing_gpu.py twice and print per-frame SER (row 0) + a checksum:
identical output twice = the kernel is run-to-run deterministic; VAEQ_LIB selects the library build."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G7_runs.npz"), allow_pickle=True)
F, N = int(g["vaele_num_frames"]), int(g["vaele_N_frame_max"])
for rep in range(2):
    SER, Var_est, var = processing("64-QAM", 2, 23, 0.0, 25, float(g["vaele_theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0",
                                   90e9, -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], dtype=np.complex64), 170,
                                   seed=int(g["vaele_seed"]), verbose=False)
    s = SER.numpy()
    conv = int(np.argmax((s < 0.1).all(0)))
    print("rep", rep, "conv frame", conv, "sha", hashlib.sha1(s.tobytes() + Var_est.numpy().tobytes()).hexdigest()[:12],
          "first SERs", np.round(s[0, :4], 4), flush=True)
