#!/usr/bin/env python3
"""Config 5 (BASELINE.json configs[4]): the script-faithful PCS x SNR grid (4 nu x 5 SNR x 3 lr x iter 5 = 300 runs, 170 frames x 10 000 symbols,
Eval_run_DP.py:24,34,67-95) through the drop-in sweep script on ONE GPU in one batch; prints wall time and per-(nu, SNR) tail statistics next to the
reference's on-grid captures (tests/golden/G13_cfg5_full_*.npz).  GPU box only.   usage: probe_config5.py [iter] [generator]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vae_equalizer_amd import Eval_run_DP as ev

NU = [0, 0.0270955, 0.0872449, 0.1222578]
SNR = [20, 22, 24, 26, 28]


def main():
    ev.nu_vec, ev.SNR_vec = NU, SNR
    ev.iter = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ev.generator = sys.argv[2] if len(sys.argv) > 2 else "hip"
    ev.base_seed = 5
    ev.savePATH = tempfile.mkdtemp() + "/"
    t0 = time.time()
    name, d = ev.main()
    wall = time.time() - t0
    S, V = d["SER"], d["Var_est"]          # [4|2, SNR, rate, nu, td, M, lr, B, fs, th, iter, frames]
    print(f"config 5 grid: {S[0, ..., 0].size} runs x {S.shape[-1]} frames in {wall:.1f} s wall (generator {ev.generator})")
    for n, nu in enumerate(NU):
        for s, snr in enumerate(SNR):
            ser = S[:, s, 0, n, 0, 0, :, 0, 0, 0, :, -30:].mean(-1).reshape(4, -1)      # [4, lr*iter] tail means
            var = V[:, s, 0, n, 0, 0, :, 0, 0, 0, :, -30:].mean(-1).reshape(2, -1)
            conv = (ser < 0.2).all(0)
            print(f"  nu={nu:<9} SNR={snr}: converged {int(conv.sum()):2d}/{conv.size}  tail SER rows {np.round(ser.mean(1), 4)}  (converged only {np.round(ser[:, conv].mean(1), 4) if conv.any() else '-'})"
                  f"  Var_est {np.round(var.mean(1), 5)}")
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for f in sorted(os.listdir(gd)):
        if f.startswith("G13_cfg5_full"):
            g = np.load(os.path.join(gd, f))
            print(f"  reference {f}: nu={float(g['nu'])} SNR={float(g['SNR'])} tail SER {np.round(g['SER'][:, -30:].mean(1), 4)} Var_est {np.round(g['Var_est'][:, -30:].mean(1), 5)}")


if __name__ == "__main__":
    main()
