"""Time the DP frame generator (vaeq_gen_dp_frame) at the bench workload: fused three-pass form vs the staged hipFFT chain (VAEQ_GEN_STAGED=1).
usage: python tools/probe_gen.py [R] [chunk] [modes, comma separated: fused,staged]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_equalizer_amd import channel as ch
from vae_equalizer_amd import shared_funcs as sfun

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sps, N = 2, 10000
h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h1", "64-QAM", "cpu", 0.0270955, sps, 25, 23)
theta = np.linspace(0, 3, R)
args = (R, N, amps, P, 23.0, h_ch, 90e9, sps, -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64), theta, "cuda:0", 1)
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fused", "staged", "fused"]
for mode in modes:
    if mode == "staged":
        os.environ["VAEQ_GEN_STAGED"] = "1"
    else:
        os.environ.pop("VAEQ_GEN_STAGED", None)
    t0 = time.perf_counter()
    ch.generate_batch_hip(*args, 0)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    ts = []
    for f in range(1, 6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ch.generate_batch_hip(*args, f)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{mode:7s} R={R} chunk={chunk}: first call {first * 1e3:8.1f} ms, then {np.median(ts) * 1e3:6.2f} ms/frame (min {min(ts) * 1e3:.2f})", flush=True)
