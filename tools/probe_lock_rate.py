#!/usr/bin/env python3
"""How often does a light-shaping config-5 run end without lock (polarisation singularity: both outputs on one polarisation), per generator form?
150 runs (nu in {0, .0270955} x 5 SNR x 3 lr x iter 5) x 170 frames per seed; fused vs staged frames (same symbols and noise, clean signal equal
to transform rounding).  Also: fused vs staged frames at run counts that exercise every runs-per-wavefront setting.   GPU box only."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vae_equalizer_amd import Eval_run_DP as ev
from vae_equalizer_amd import channel as ch
from vae_equalizer_amd import shared_funcs as sfun


def frames(R, frame, staged):
    if staged:
        os.environ["VAEQ_GEN_STAGED"] = "1"
    else:
        os.environ.pop("VAEQ_GEN_STAGED", None)
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", "64-QAM", "cpu", 0.0270955, 2, 25, 23)
    theta = np.linspace(0, 40, R)
    return ch.generate_batch_hip(R, 10000, amps, P, np.linspace(18, 30, R).astype(np.float32), h_ch, 90e9, 2, -26e-24, 0.1e-12 * np.sqrt(1000),
                                 np.array([0.0314, 0.0314], np.complex64), theta, "cuda:0", 3, frame)


for R in (300, 1500, 4100):
    for f in (0, 7):
        a, da = frames(R, f, False)
        b, db = frames(R, f, True)
        d = (a - b).abs().amax(dim=(1, 2, 3)) / b.abs().amax()
        print(f"R={R} frame {f}: max |fused - staged| / max|rx| = {float(d.max()):.2e} (run {int(d.argmax())}), TX reference equal: {torch.equal(da, db)}", flush=True)
    del a, b
os.environ.pop("VAEQ_GEN_STAGED", None)

NU, SNR = [0, 0.0270955], [20, 22, 24, 26, 28]
ev.nu_vec, ev.SNR_vec, ev.generator = NU, SNR, "hip"
ev.savePATH = tempfile.mkdtemp() + "/"
for mode in ("fused", "staged"):
    if mode == "staged":
        os.environ["VAEQ_GEN_STAGED"] = "1"
    else:
        os.environ.pop("VAEQ_GEN_STAGED", None)
    tot = bad = 0
    for seed in range(1, 9):
        ev.base_seed = seed
        name, d = ev.main()
        S = d["SER"]                                   # [4, SNR, rate, nu, td, M, lr, B, fs, th, iter, frames]
        t = S[..., -30:].mean(-1)
        un = np.argwhere((t > 0.2).any(0))
        tot += t[0].size
        bad += len(un)
        for u in un:
            tr = S[(slice(None),) + tuple(u)]          # [4, frames]
            first = [int(np.argmax(tr[r] < 0.2)) if (tr[r] < 0.2).any() else -1 for r in range(4)]
            print(f"  {mode} seed {seed}: run SNR={SNR[u[0]]} nu={NU[u[2]]} lr#{u[5]} iter {u[9]}: tail SER rows {np.round(t[(slice(None),) + tuple(u)], 3)}, first frame below 0.2 per row {first}", flush=True)
    print(f"{mode}: {bad} of {tot} light-shaping runs end without lock", flush=True)
