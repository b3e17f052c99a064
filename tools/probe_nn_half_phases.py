#!/usr/bin/env python3
"""Per-phase wall-clock ticks (100 MHz) of the last step of run 0 of nn_train_half_kernel, from a library built with -DVAEQ_NN_HALF_STAMPS
(hipcc ... -DVAEQ_NN_HALF_STAMPS -> gpurun_variants/libvaeq_nnhprof.so):  VAEQ_LIB=... python tools/probe_nn_half_phases.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VAEQ_NN_HALF"] = "1"
import torch
from vae_equalizer_amd.engine import NNEngine
from vae_equalizer_amd.func_VAENN_MQAM import vaenn_tables
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = vaenn_tables("64-QAM", "h1", 2)
eng = NNEngine(R, 25, 25, 3, t["amps"], "cuda:0", 2)
eng.init_parameters()
rx = 0.5 * torch.randn(R, 2, 13 * 600, device="cuda:0")
for _ in range(2):
    out = eng.train(rx, 300, 13, 4e-3)
torch.cuda.synchronize()
ticks = out["loss"][0, :13].cpu().numpy()
names = ["forward (2 x fc1, fc2)", "softmax", "residual + C", "dh + dq", "gw2 half 1", "convT half 1", "gw1 half 1", "-", "fc1 recompute half 0",
         "gw2 half 0", "convT half 0", "gw1 half 0", "adam + transposes"]
for n, v in zip(names, ticks):
    print(f"{n:26s} {v / 100:8.2f} us")
print(f"R={R} sum {ticks.sum() / 100:.2f} us")
