#!/usr/bin/env python3
"""Per-frame wall time of the whole drop-in pipeline (on-device channel simulator -> training kernel -> epilogue) for R runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd import channel as ch, epilogue as epi, shared_funcs as sfun
from vae_equalizer_amd.engine import DPEngine, dp_epilogue, dp_epilogue_compact

dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C = bench.CFG
t = sfun.qam_tables(C["mod"], C["nu"]); h_ch = sfun.upsampled_channel(C["channel"], 2)
var = t["pow_mean"] / 10 ** 2.3 / 2
eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
amp = torch.tensor(t["amps"], dtype=torch.float32, device=dev)
nu = torch.zeros(R, device=dev); varr = torch.full((R, 2), var, device=dev)
gen = torch.Generator(device=dev).manual_seed(0)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = sync()
    rx, data = ch.generate_batch_hip(R, 10000, t["amps"], t["P"], 23.0, h_ch, 90e9, 2, C["tau_cd"], C["tau_pmd"], C["phiIQ"], 0.3, dev, 1, it)
    t1 = sync()
    compact = len(sys.argv) > 2 and sys.argv[2] == "compact"
    out = eng.train(rx, 100, 100, 2.5e-3, want_q=not compact, want_compact=compact)
    t2 = sync()
    if compact:
        res = dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100)
    else:
        res = dp_epilogue(out["q"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100)
    ser = res["SER"].cpu()
    t3 = sync()
    print(f"R={R} frame {it}: generate {1e3*(t1-t0):8.1f} ms | train {1e3*(t2-t1):8.1f} ms | epilogue {1e3*(t3-t2):8.1f} ms", flush=True)
