#!/usr/bin/env python3
"""Shader-clock cycles per phase of the last step of one wave of the DP wave kernel under full load (library built with the phase
stamps: tools/snap_variant.sh stamps -DVAEQ_PHASE_STAMPS -> gpurun_variants/libvaeq_stamps.so).  VAEQ_LIB=... python tools/probe_dp_phases.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import DPEngine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
eng = DPEngine(R, 25, amp, np.full(8, 1 / 8, np.float32), [0.0025, 0.0025], 0.0, "cuda:0", 2, 1)
rx = 0.4 * torch.randn(R, 1, 2, 2, 20000, device="cuda:0")
for _ in range(2):
    out = eng.train(rx, 100, 100, 2.5e-3)
torch.cuda.synchronize()
t = out["loss"].reshape(-1)[:11].cpu().numpy()
names = ["P0 window->LDS", "P1 FIR", "P1 y stores", "P2 demap + q stores", "P2 scan, VS", "P3 D conv", "P3 e, sums, loss, PSh", "P4a dh + Adam(h)",
         "P4b dU, dy", "P4b h, gy -> LDS", "P5 dW + Adam(W)"]
for n, v in zip(names, t):
    print(f"{n:20s} {v:9.0f} cycles  {100 * v / t.sum():5.1f} %")
print("sum", t.sum(), "cycles per step")
x = out["loss"].reshape(-1)[16:23].cpu().numpy()
for n, v in zip(["dh: tap loop", "dh: combine + cross-half shuffle", "dh: Adam(h) + window prefetch issue", "(dU .. gy -> LDS)", "dw: tap loop", "dw: combine + cross-half shuffle", "dw: Adam(w)"], x):
    print(f"   {n:40s} {v:9.0f} cycles")
