#!/usr/bin/env python3
"""The compact epilogue alone at R runs x 10 000 symbols: dp_epilogue_compact_kernel with / without the TX level cache (default / VAEQ_EPI_NOTXC=1) vs the re-reading dp_epilogue_kernel (VAEQ_EPI_REREAD=1), alternating.
tools/probe_epilogue.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd import channel as ch, shared_funcs as sfun
from vae_equalizer_amd.engine import DPEngine, dp_epilogue_compact
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = bench.CFG
t = sfun.qam_tables(C["mod"], C["nu"]); h_ch = sfun.upsampled_channel(C["channel"], 2)
var = t["pow_mean"] / 10 ** 2.3 / 2
eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
amp = torch.tensor(t["amps"], dtype=torch.float32, device=dev)
nu = torch.zeros(R, device=dev); varr = torch.full((R, 2), var, device=dev)
rx, data = ch.generate_batch_hip(R, 10000, t["amps"], t["P"], 23.0, h_ch, 90e9, 2, C["tau_cd"], C["tau_pmd"], C["phiIQ"], 0.3, dev, 1, 0)
out = eng.train(rx, 100, 100, 2.5e-3, want_q=False, want_compact=True)
res = {}
for rep in range(3):
    for mode in ("txc", "notxc", "reread"):
        os.environ.pop("VAEQ_EPI_REREAD", None); os.environ.pop("VAEQ_EPI_NOTXC", None)
        if mode == "reread":
            os.environ["VAEQ_EPI_REREAD"] = "1"
        if mode == "notxc":
            os.environ["VAEQ_EPI_NOTXC"] = "1"
        f = lambda: res.__setitem__(mode, dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100))
        f()
        ms = np.median(bench._event_ms(f, 5))
        print(f"R={R} epilogue {mode:7s}: {ms:.3f} ms", flush=True)
print("identical:", all(torch.equal(res["txc"][k], res["reread"][k]) and torch.equal(res["notxc"][k], res["reread"][k]) for k in res["txc"]))
