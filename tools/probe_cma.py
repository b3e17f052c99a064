#!/usr/bin/env python3
"""Throughput of the constant-modulus baseline kernel (row f4) and the phase estimation at R runs x 10000 symbols per frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import cma, cpe
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda:0"
rx = 0.5 * torch.randn(R, 2, 2, 20000, device=dev)
for mode, lr in (("CMA", 1e-3), ("CMAbatch", 1e-5), ("CMAflex", 1e-6)):
    h = torch.zeros(R, 2, 2, 2, 25, device=dev); h[:, 0, 0, 0, 12] = 1; h[:, 1, 1, 0, 12] = 1
    cma(rx, h, lr, 2, mode, 100, 10); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out, _ = cma(rx, h, lr, 2, mode, 100, 10, want_e=False); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{mode:9s} R={R}: {ms:8.2f} ms per frame of 10000 symbols -> {R * 10000 / ms / 1e6:.2f} G DP-symbols/s", flush=True)
y = out[..., 10:-10].contiguous()
cpe(y); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); cpe(y); e1.record(); torch.cuda.synchronize()
print(f"CPE       R={R}: {e0.elapsed_time(e1):8.2f} ms", flush=True)
