#!/usr/bin/env python3
"""bench.py's parity gate quantities (GPU vs C oracle, free run from the Dirac start on the bench's own frames) as a function of the number of
free-running steps: where fp32 chaos lifts the tap deviation above the gate's tolerance.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd.engine import DPEngine
dev = torch.device("cuda", 0)
R = 256
frames, t, data0 = bench.make_frames(1, R, dev, seed=1000, with_data=True)
var = t["pow_mean"] / 10 ** (bench.CFG["SNR"] / 10) / 2
lr = np.array([bench.CFG["lr_optim_vec"][i % 3] for i in range(R)], np.float32)
rx_np = frames[0][:, 0].cpu().numpy()
B, M, sps = 100, 25, 2
for steps in (1, 2, 5, 10, 15, 20, 30):
    eng = DPEngine(R, M, t["amps"], t["P"], [var, var], t["nu_sc"], dev, sps, 0)
    g = eng.train(frames[0], B, steps, torch.tensor(lr, device=dev), want_q=False)
    torch.cuda.synchronize()
    o = bench._oracle_train(rx_np[..., :steps * B * sps], t, var, lr, steps, bench.host_cores())
    gl = g["loss"][:, 0].cpu().numpy()
    dl = np.abs(gl - o["loss"]) / np.abs(o["loss"])
    dW = np.abs(eng.W.cpu().numpy() - o["W"]).reshape(R, -1).max(1)
    dh = np.abs(eng.h.cpu().numpy() - o["h"]).reshape(R, -1).max(1)
    dt = np.maximum(dW, dh)
    print(f"steps {steps:3d}: ELBO rel max {dl.max():.2e} (median run {np.median(dl.max(1)):.2e})   taps abs max over {R} runs {dt.max():.2e}, median run {np.median(dt):.2e}, "
          f"first 64 runs {dt[:64].max():.2e}", flush=True)
