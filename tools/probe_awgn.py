#!/usr/bin/env python3
"""AWGN VAE-LE training kernel throughput (config-2 shape: 64-QAM, B=350 [argv 4], M=25) vs number of runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import AWGNEngine
dev = "cuda:0"
amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
steps, B = int(sys.argv[2]) if len(sys.argv) > 2 else 30, int(sys.argv[4]) if len(sys.argv) > 4 else 350
for th in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"])]:
    for R in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,1024,2048,4096".split(","))]:
        rx = 0.4 * torch.randn(R, 2, steps * B * 2, device=dev)
        eng = AWGNEngine(R, 25, amp, np.full(8, 1 / 8, np.float32), 0.56, 0.004, dev, 2, th)
        for _ in range(2):
            eng.train(rx, B, steps, 5e-3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.train(rx, B, steps, 5e-3)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"threads={th} B={B} R={R:6d} {ms:8.3f} ms {ms*1e3/steps:8.2f} us/step {R*steps*B/ms/1e6:8.3f} G sym/s", flush=True)
