#!/usr/bin/env python3
"""Per-epoch cost of the AWGN sweep pipeline (config 2 shape) at R runs: generator (train + validation frames), training launch,
fused validation launch.  python tools/probe_awgn_pipeline.py [R]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd import channel as ch
from vae_equalizer_amd.engine import AWGNEngine
from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev, sps = "cuda:0", 2
t = awgn_tables("64-QAM", 0.0270955, 24, "h1", sps)
eng = AWGNEngine(R, 25, t["amps"], np.tile(t["P"], (R, 1)), t["amp_mean"], t["var"], dev, sps)
snr = np.full(R, 24, np.float32)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()                                   # result dropped at once: the caching allocator reuses the buffers like a sweep loop does
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, fn()


tg1, (rx, _) = timed(lambda: ch.generate_awgn_batch_hip(R, 1200, t["amps"], t["P"], snr, t["h_channel"], sps, dev, 1, 0))
tg2, (rxv, dv) = timed(lambda: ch.generate_awgn_batch_hip(R, 15000, t["amps"], t["P"], snr, t["h_channel"], sps, dev, 1, 1))
tt, _ = timed(lambda: eng.train(rx, 350, 3, 5e-3))
tv, _ = timed(lambda: eng.validate(rxv, dv, 21))
tf, _ = timed(lambda: eng.forward(rxv))
del rxv, dv
tc, clean = timed(lambda: ch.generate_awgn_clean_batch_hip(R, 15000, t["amps"], t["P"], snr, t["h_channel"], sps, dev, 1, 1))
tvc, _ = timed(lambda: eng.validate_clean(clean, 21))
print(f"R={R}: gen(1200) {tg1:.3f} ms  gen(15000) {tg2:.3f} ms  train(3x350) {tt:.3f} ms  validate(15000, fused) {tv:.3f} ms  "
      f"[forward with q: {tf:.3f} ms]  epoch total {tg1 + tg2 + tt + tv:.3f} ms -> {R / (tg1 + tg2 + tt + tv) * 1e3:.0f} run-epochs/s", flush=True)
print(f"R={R}: validation frame clean + noise on load: gen_clean(15000) {tc:.3f} ms  validate_gen {tvc:.3f} ms = {tc + tvc:.3f} ms "
      f"(two-step: {tg2 + tv:.3f} ms)  epoch total {tg1 + tt + tc + tvc:.3f} ms", flush=True)
