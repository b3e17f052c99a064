#!/usr/bin/env python3
"""Training launches of a SMALL sweep (R runs, one wavefront each): G frames per launch vs one -- what a launch boundary costs between two dependent
launches.  tools/probe_frames_per_launch.py [R] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd import shared_funcs as sfun
from vae_equalizer_amd.engine import DPEngine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
C = bench.CFG
t = sfun.qam_tables(C["mod"], C["nu"])
var = t["pow_mean"] / 10 ** 2.3 / 2
rx = 0.3 * torch.randn(R, F, 2, 2, 20000, device=dev)
for G in (1, 2, 4, 8, 16, 1, 4):
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
    def go():
        for f in range(0, F, G):
            eng.train(rx[:, f:f + G].contiguous() if G < F else rx, 100, 100, 2.5e-3, want_q=False, want_compact=True)
    parts = [rx[:, f:f + G].contiguous() for f in range(0, F, G)]
    def go2():
        for p in parts:
            eng.train(p, 100, 100, 2.5e-3, want_q=False, want_compact=True)
    go2(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go2(); e1.record(); torch.cuda.synchronize()
    print(f"R={R}: {F} frames as launches of {G:2d}: {e0.elapsed_time(e1) / F:.3f} ms per frame", flush=True)
