#!/usr/bin/env python3
"""Per-phase wall-clock times of the DP epilogue kernel for run 0 (tools/build_phase_probe.sh epilogue -> gpurun_variants/libvaeq_epilogueprof.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import dp_epilogue_compact
R, N, dev = 8192, 10000, "cuda:0"
amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
eq = torch.randn(R, 2, N, device=dev)
dec = torch.randint(0, 8, (R, 2, 2, N), device=dev, dtype=torch.int8)
y = torch.randn(R, 2, 2, N, device=dev)
data = torch.from_numpy(amp)[torch.randint(0, 8, (R, 2, 2, N))].to(torch.float16).to(dev)
for _ in range(2):
    res = dp_epilogue_compact(eq, dec, y, data, amp, 0.0, 0.01, 100)
torch.cuda.synchronize()
t = eq[0, 0, :8].cpu().numpy() / 100
print("correlate q", t[0], "| SER q", t[2], "| correlate y", t[4], "| radius", t[5], "| SER y", t[6], "us   (others:", t[1], t[3], ")")
