#!/usr/bin/env python3
"""VAE-NN (row f3) kernel times at R runs, the sweep script's shape (64-QAM, k1 = 25, k2 = 3, M = 25, batch_len 300):
training launch of 13 minibatches (train_len 4000), fused validation on 15000 symbols, generator.  python tools/probe_nn.py [R] [mod] [net: Net | Net_BN]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd import channel as ch
from vae_equalizer_amd.engine import NNEngine
from vae_equalizer_amd.func_VAENN_MQAM import vaenn_tables
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev, sps = "cuda:0", 2
mod = sys.argv[2] if len(sys.argv) > 2 else "64-QAM"
bn = len(sys.argv) > 3 and sys.argv[3] == "Net_BN"
t = vaenn_tables(mod, "h1", sps)
eng = NNEngine(R, 25, 25, 3, t["amps"], dev, sps, batch_norm=bn)
eng.init_parameters()
nl = len(t["amps"])
P, sig = np.full(nl, 1 / nl), np.full(R, np.sqrt(0.5) / 10 ** (24 / 20), np.float32)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, fn()


tg, (rx, _) = timed(lambda: ch.generate_awgn_batch_hip(R, 4000, t["amps"], P, 24.0, t["h_channel"], sps, dev, 1, 0, sigma_fixed=sig))
tg2, (rxv, dv) = timed(lambda: ch.generate_awgn_batch_hip(R, 15000, t["amps"], P, 24.0, t["h_channel"], sps, dev, 1, 1, sigma_fixed=sig))
tt, _ = timed(lambda: eng.train(rx, 300, 13, 4e-3))
tv, _ = timed(lambda: eng.validate(rxv, dv, 21))
macs = 13 * (600 * 16 * 50 * 3 + 300 * 16 * 48 * 3) + 0
net = "Net_BN" if bn else "Net"
print(f"R={R} {mod} {net}: gen {tg:.3f} + {tg2:.3f} ms | train(13 x 300) {tt:.3f} ms = {tt * 1e3 / 13:.1f} us/step, {R * 13 * 300 / tt / 1e6:.3f} G sym/s, "
      f"{2 * R * macs / tt / 1e9:.2f} TFLOP/s | validate(15000) {tv:.3f} ms, {2 * R * 15000 * (2 * 16 * 50 + 16 * 48) / tv / 1e9:.2f} TFLOP/s", flush=True)
