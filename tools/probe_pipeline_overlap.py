#!/usr/bin/env python3
"""Does running the channel simulator of frame f+1 and the epilogue of frame f on side streams beside / between the training launches shorten the
per-frame time of the drop-in pipeline (8192 runs, compact mode)?  Serial order vs three streams, same kernels, identical results.  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd import channel as ch, shared_funcs as sfun
from vae_equalizer_amd.engine import DPEngine, dp_epilogue_compact

dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C = bench.CFG
t = sfun.qam_tables(C["mod"], C["nu"]); h_ch = sfun.upsampled_channel(C["channel"], 2)
var = t["pow_mean"] / 10 ** 2.3 / 2
amp = torch.tensor(t["amps"], dtype=torch.float32, device=dev)
nu = torch.zeros(R, device=dev); varr = torch.full((R, 2), var, device=dev)

def gen(f):
    return ch.generate_batch_hip(R, 10000, t["amps"], t["P"], 23.0, h_ch, 90e9, 2, C["tau_cd"], C["tau_pmd"], C["phiIQ"], 0.3 + 0.06 * np.pi * f, dev, 1, f)

def serial():
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
    sers = []
    for f in range(F):
        rx, data = gen(f)
        out = eng.train(rx, 100, 100, 2.5e-3, want_q=False, want_compact=True)
        sers.append(dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100)["SER"])
    return torch.stack(sers)

def overlapped():
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
    main = torch.cuda.current_stream(dev)
    gs, es = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    sers, keep = [], []
    gs.wait_stream(main)
    with torch.cuda.stream(gs):
        nxt = gen(0); g_done = torch.cuda.Event(); g_done.record(gs)
    for f in range(F):
        main.wait_event(g_done)
        rx, data = nxt
        out = eng.train(rx, 100, 100, 2.5e-3, want_q=False, want_compact=True)
        t_done = torch.cuda.Event(); t_done.record(main)
        if f + 1 < F:
            with torch.cuda.stream(gs):                        # the next frame's samples: independent of the training state
                nxt = gen(f + 1); g_done = torch.cuda.Event(); g_done.record(gs)
        es.wait_event(t_done)
        with torch.cuda.stream(es):
            sers.append(dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100)["SER"])
        keep.append((out, rx, data))                           # alive until the side streams are done with them
        if len(keep) > 2: keep.pop(0)
    main.wait_stream(es); main.wait_stream(gs)
    return torch.stack(sers)

for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = fn()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{name:11s} R={R} {F} frames: {1e3 * el / F:7.2f} ms per frame  {R * 10000 * F / el / 1e9:.3f} G DP symbols/s   SER checksum {float(S.double().sum()):.6f}", flush=True)
