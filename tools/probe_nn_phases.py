#!/usr/bin/env python3
"""Per-phase wall-clock ticks (100 MHz) of the last VAE-NN training step of run 0, from a library built with -DVAEQ_NN_STAMPS
(tools/build_phase_probe.sh nn -> gpurun_variants/libvaeq_nnprof.so): VAEQ_LIB=... python tools/probe_nn_phases.py [R] [Net | Net_BN]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import NNEngine
from vae_equalizer_amd.func_VAENN_MQAM import vaenn_tables
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bn = len(sys.argv) > 2 and sys.argv[2] == "Net_BN"
t = vaenn_tables("64-QAM", "h1", 2)
eng = NNEngine(R, 25, 25, 3, t["amps"], "cuda:0", 2, batch_norm=bn)
eng.init_parameters()
rx = 0.5 * torch.randn(R, 2, 13 * 600, device="cuda:0")
for _ in range(2):
    out = eng.train(rx, 300, 13, 4e-3, debug_grads=True)
torch.cuda.synchronize()
ticks = out["loss"][0, :10].cpu().numpy()
names = ["P0 minibatch -> LDS", "P1/P2 fc1+fc2", "P3 softmax", "P4 residual + C", "P5 dh", "P6 dq", "P7a gw2", "P7b gz", "P8 gw1", "P9 adam+transpose"]
for n, v in zip(names, ticks):
    print(f"{n:20s} {v / 100:8.2f} us")
print("R=%d sum %.2f us" % (R, ticks.sum() / 100))
