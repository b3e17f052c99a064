#!/usr/bin/env python3
"""Small sweeps with the stages of a frame on DISJOINT compute units (hipExtStreamCreateWithCUMask): training launches on one set, channel model and
epilogue on the rest -- does keeping the side streams' waves off the SIMDs of the (latency-bound, one wave per run) training kernel lower the per-frame
time of run_dp_batch below what three unmasked streams reach?   tools/probe_cu_mask.py [frames] [train CUs, comma-separated]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd import dp_runs
from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
splits = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["160", "192"])]
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(lo, hi):
    words = (NCU + 31) // 32
    m = (C.c_uint32 * words)()
    for b in range(lo, hi):
        m[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, m)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, dev)


NU, SNR = [0, 0.0270955, 0.0872449, 0.1222578], [20, 22, 24, 26, 28]
runs = [DPRun(s, nu, 0.06 * np.pi, np.pi / 10, lr, 90e9) for nu in NU for lr in (2.5e-3, 2e-3, 3e-3) for s in SNR for i in range(5)]
runs[0].seed = 4321


def once(tag, main=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if main is None:
        r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 10000, F, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64), 170, generator="hip")
    else:
        with torch.cuda.stream(main):
            r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 10000, F, 10, "h0", -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], np.complex64), 170, generator="hip")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{len(runs):4d} runs x {F} frames, {tag:34s}: {1e3 * (t1 - t0) / F:.3f} ms per frame   SER checksum {float(r['SER'].double().sum()):.6f}", flush=True)


print(f"{NCU} compute units", flush=True)
plain = dp_runs._side_streams(dev)
for rep in range(2):
    dp_runs._SIDE[str(dev)] = plain
    once("three plain streams")
    for k in splits:
        main = masked_stream(0, k)
        dp_runs._SIDE[str(dev)] = (masked_stream(k, NCU), masked_stream(k, NCU))
        once(f"train on {k} CUs, sides on {NCU - k}", main)
        dp_runs._SIDE[str(dev)] = (masked_stream(0, NCU), masked_stream(0, NCU))
        once(f"all-CU masks (control)", masked_stream(0, NCU))
