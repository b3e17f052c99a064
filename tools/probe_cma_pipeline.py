#!/usr/bin/env python3
"""Per-frame cost of the constant-modulus baselines' drop-in pipeline (row f4) at R runs: generator, CMA kernel, two-stage epilogue
(phase estimation + constellation SER + soft demapper SER).   python tools/probe_cma_pipeline.py [R] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd import channel as ch, shared_funcs as sfun
from vae_equalizer_amd.engine import cma
from vae_equalizer_amd.cma_runs import cma_frame_epilogue
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mode = sys.argv[2] if len(sys.argv) > 2 else "CMA"
dev, sps, N = "cuda:0", 2, 10000
t = sfun.qam_tables("64-QAM", 0.0)
h_ch = sfun.upsampled_channel("h0", sps)
amp = torch.tensor(t["amps"], dtype=torch.float32, device=dev)
var = torch.full((R, 2), float(t["pow_mean"] / 10 ** 2.3 / 2), device=dev)
nu = torch.zeros(R, device=dev)
h = torch.zeros(R, 2, 2, 2, 25, device=dev); h[:, 0, 0, 0, 12] = 1; h[:, 1, 1, 0, 12] = 1
lr = np.full(R, 1e-3, np.float32)


def ev(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


tg, (rx, data) = ev(lambda: ch.generate_batch_hip(R, N, t["amps"], t["P"], 23.0, h_ch, 90e9, sps, -26e-24, 0.1e-12 * np.sqrt(1000),
                                                  np.array([0.0314, 0.0314], np.complex64), np.zeros(R), dev, 1, 0))
tk, (out_const, _) = ev(lambda: cma(rx, h, lr, sps, mode, 100, 10, 1.0, want_e=False))
te, res = ev(lambda: cma_frame_epilogue(out_const, data, amp, nu, var))
print(f"R={R} {mode}: generate {tg:.2f} ms | kernel {tk:.2f} ms | epilogue {te:.2f} ms  -> {R * N / (tg + tk + te) / 1e6:.3f} G DP symbols/s end to end; SER {res['SER'].mean(0).cpu().numpy().round(3)}", flush=True)
