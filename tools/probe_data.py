#!/usr/bin/env python3
"""Does the kernel time depend on the data?  random Gaussian samples vs simulated channel frames, wave vs generic kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd.engine import DPEngine

dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
frames, t = bench.make_frames(2, R, dev, 1)
var = t["pow_mean"] / 10 ** 2.3 / 2
g = torch.Generator(device=dev).manual_seed(0)
rnd = 0.4 * torch.randn(R, 1, 2, 2, 20000, device=dev, generator=g)
for name, rx in (("random", rnd), ("channel", frames[0])):
    for th in (1, 256):
        eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2, th)
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = eng.train(rx, 100, 100, 2.5e-3)
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:8s} threads={th:3d} call {it}: {e0.elapsed_time(e1):8.3f} ms  loss[-1]={float(out['loss'][0,0,-1]):.2f} finite={bool(torch.isfinite(out['loss']).all())}", flush=True)
print("--- bench-like flow: cycling frames, per-run lr tensor")
lr = torch.tensor(np.array([bench.CFG["lr_optim_vec"][i % 3] for i in range(R)], np.float32), device=dev)
for th in (1,):
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2, th)
    for it in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = eng.train(frames[it % 2], 100, 100, lr)
        e1.record()
        torch.cuda.synchronize()
        print(f"threads={th:3d} call {it}: {e0.elapsed_time(e1):8.3f} ms  loss[-1]={float(out['loss'][0,0,-1]):.2f} min|q|>0: {float(out['q'][out['q']>0].min()):.3e}", flush=True)
