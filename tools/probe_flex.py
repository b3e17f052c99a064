#!/usr/bin/env python3
"""Kernel time of one VAEflex frame (BASELINE config 4: window 100 symbols, stride 10, centre 10 kept -> 990 window steps per 10 000-symbol frame,
func_VAEflex_DP_MQAM_shaping.py:59-70) for R runs.  GPU box only.   usage: probe_flex.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd import _native as nat
from vae_equalizer_amd.engine import DPEngine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B, fs, N = 100, 10, 10000
steps = (N - B) // fs
amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
rx = 0.4 * torch.randn(R, 1, 2, 2, 2 * N, device="cuda:0")
eng = DPEngine(R, 25, amp, np.full(8, 1 / 8, np.float32), [0.0025, 0.0025], 0.0, "cuda:0", 2)
go = lambda: eng.train(rx, B, steps, 2.5e-3, stride=fs, keep_off=(B - fs) // 2, keep_len=fs)
go(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2): go()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
print(f"{nat.last_kernel()}  R={R} steps={steps}: {ms:.2f} ms  {ms * 1e3 / steps:.2f} us/step  {R * steps * fs / ms / 1e6:.4f} G output sym/s", flush=True)
