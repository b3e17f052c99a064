#!/usr/bin/env python3
"""Where the wall time of the config-5 grid (300 runs x 170 frames through Eval_run_DP.main(), one GPU) goes: cProfile of a second, warm invocation."""
import os, sys, time, tempfile, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_equalizer_amd import Eval_run_DP as ev
ev.nu_vec, ev.SNR_vec = [0, 0.0270955, 0.0872449, 0.1222578], [20, 22, 24, 26, 28]
ev.generator, ev.base_seed, ev.savePATH = "hip", 5, tempfile.mkdtemp() + "/"
t0 = time.time(); ev.main(); t1 = time.time()
pr = cProfile.Profile(); pr.enable(); ev.main(); pr.disable(); t2 = time.time()
print(f"cold run {t1 - t0:.2f} s, warm run {t2 - t1:.2f} s ({(t2 - t1) / 170 * 1e3:.1f} ms per frame)")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
