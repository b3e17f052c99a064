#!/usr/bin/env python3
"""Kernel time of one frame (100 minibatch steps of B=100 [argv 4], M=25 [argv 5], 64-QAM) vs number of runs, per kernel variant.
Shows residency (time is flat while all runs are co-resident) and per-step latency.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import DPEngine

def main():
    dev = "cuda:0"
    amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
    variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "256"])]
    Rs = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "64,256,512,1024,2048,3072,4096,6144".split(","))]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    M = int(sys.argv[5]) if len(sys.argv) > 5 else 25
    g = torch.Generator(device=dev).manual_seed(0)
    for th in variants:
        for R in Rs:
            rx = 0.4 * torch.randn(R, 1, 2, 2, steps * 2 * B, device=dev, generator=g)
            eng = DPEngine(R, M, amp, np.full(8, 1 / 8, np.float32), [0.0025, 0.0025], 0.0, dev, 2, th)
            for _ in range(2):
                out = eng.train(rx, B, steps, 2.5e-3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                out = eng.train(rx, B, steps, 2.5e-3)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"threads={th:4d} B={B} R={R:6d}  {ms:9.3f} ms  {ms * 1e3 / steps:8.2f} us/step  {R * steps * B / ms / 1e6:9.3f} G sym/s", flush=True)
            del out, eng, rx

if __name__ == "__main__":
    main()
