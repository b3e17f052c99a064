#!/usr/bin/env python3
"""Drop-in pipeline at 8192 runs: can the two stages AROUND the training launch share the chip with EACH OTHER?

The training kernel owns the register file for its whole launch (nothing runs beside it: pipeline_overlap_probe, pipeline_chunked_overlap).  But the
epilogue of frame f is bound by vector issue (74 % VALU-busy) and the channel simulator of frame f + 1 mostly by memory (its FFT and finish passes move
10.7 GB at 4-5 TB/s; only its first pass is issue-bound).  Order per frame here:  train(f)  ->  { epilogue(f)  ||  generate(f + 1) }  ->  train(f + 1),
the generator optionally as K run-chunks round-robin on several streams so that its own issue-bound and memory-bound passes overlap too.
(Chunks use their own Philox keys here: a timing probe; the library call has no run offset yet.)  GPU box only.

usage: probe_pipeline_phased.py [R=8192] [F=8]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd import channel as ch, shared_funcs as sfun, _native as nat
from vae_equalizer_amd.engine import DPEngine, dp_epilogue_compact

dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
CFG = bench.CFG
t = sfun.qam_tables(CFG["mod"], CFG["nu"]); h_ch = sfun.upsampled_channel(CFG["channel"], 2)
var = t["pow_mean"] / 10 ** 2.3 / 2
amp = torch.tensor(t["amps"], dtype=torch.float32, device=dev)
nu = torch.zeros(R, device=dev); varr = torch.full((R, 2), var, device=dev)
N, sps = 10000, 2
geo = ch.dp_frame_geometry(N, h_ch, sps)
n = len(t["amps"])
amp_t = ch._dev_const(t["amps"], torch.float32, dev)
cdf = ch._cdf_dev(t["P"], R, n, dev)
g_t = ch._dev_const(np.stack([geo["g"].real, geo["g"].imag], -1), torch.float32, dev)
snr = ch._dev_const(np.broadcast_to(np.asarray(23.0, np.float32), (R,)), torch.float32, dev)
e = np.exp(-1j * np.asarray(CFG["phiIQ"], dtype=np.complex128))
Lrow = ch.padded_row_len(geo["Ls"] + 64)
L = nat.lib()
NPW = L.vaeq_gen_dp_power_parts(Lrow)
SIDE = [torch.cuda.Stream(dev) for _ in range(4)]


class Frame:
    def __init__(self):
        self.rx = torch.empty(R, 2, 2, sps * N, dtype=torch.float32, device=dev)
        self.data = torch.empty(R, 2, 2, N, dtype=torch.float16, device=dev)
        self.sigma = torch.empty(R, dtype=torch.float32, device=dev)
        self.sig = torch.empty(R, 2, Lrow, 2, dtype=torch.float32, device=dev)
        self.pw = torch.empty(R, NPW, dtype=torch.float32, device=dev)


def gen_into(fr, f, K, streams):
    """vaeq_gen_dp_frame for K chunks of runs, chunk c on streams[c % len(streams)] (None = the current stream)."""
    th = ch._dev_const(np.broadcast_to(np.asarray(0.3 + 0.06 * np.pi * f, np.float32), (R,)), torch.float32, dev)
    cur = torch.cuda.current_stream(dev)
    step = (R + K - 1) // K
    for c, r0 in enumerate(range(0, R, step)):
        r1 = min(R, r0 + step)
        s = streams[c % len(streams)] if streams else cur
        if s is not cur:
            s.wait_stream(cur)
        with torch.cuda.stream(s):
            nat.check(L.vaeq_gen_dp_frame(r1 - r0, N, geo["N_conv"], sps, n, geo["Lg"], geo["Ls"], Lrow, geo["ref_offset"], nat.ptr(amp_t),
                                          nat.ptr(cdf[r0:r1]), nat.ptr(g_t), nat.ptr(snr[r0:r1]), nat.ptr(th[r0:r1]), 90e9 * sps, float(CFG["tau_cd"]),
                                          float(CFG["tau_pmd"]), float(e[0].real), float(e[0].imag), float(e[1].real), float(e[1].imag),
                                          C.c_uint64(ch._mix_seed(1, r0)), C.c_uint32(f), nat.ptr(fr.sig[r0:r1]), nat.ptr(fr.pw[r0:r1]), nat.ptr(fr.rx[r0:r1]),
                                          nat.ptr(fr.data[r0:r1], torch.float16), nat.ptr(fr.sigma[r0:r1]), nat.current_stream(dev)), "vaeq_gen_dp_frame")
    if streams:
        for s in streams:
            cur.wait_stream(s)


def epi(out, data):
    return dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100)["SER"]


FR = [Frame(), Frame()]


def pipeline(mode, K, ns):
    """mode 'serial': generate, train, epilogue on one stream.  'phased': train(f), then epilogue(f) on a side stream beside generate(f + 1)."""
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
    cur = torch.cuda.current_stream(dev)
    streams = SIDE[:ns] if ns else None
    sers, keep = [], []
    if mode == "serial":
        for f in range(F):
            fr = FR[f & 1]
            gen_into(fr, f, K, streams)
            out = eng.train(fr.rx, 100, 100, 2.5e-3, want_q=False, want_compact=True)
            sers.append(epi(out, fr.data)); keep.append(out)
        return torch.stack(sers)
    es = SIDE[3]
    gen_into(FR[0], 0, K, streams)
    for f in range(F):
        fr = FR[f & 1]
        out = eng.train(fr.rx, 100, 100, 2.5e-3, want_q=False, want_compact=True)
        keep.append(out)
        es.wait_stream(cur)
        with torch.cuda.stream(es):
            sers.append(epi(out, fr.data))
        if f + 1 < F:
            gen_into(FR[(f + 1) & 1], f + 1, K, streams if streams else [SIDE[0]])
        cur.wait_stream(es)
    return torch.stack(sers)


def timed(name, fn, per=F):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    S = fn()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    extra = f"   SER checksum {float(S.double().sum()):.6f}" if S is not None else ""
    print(f"{name:52s} {1e3 * el / per:7.2f} ms per frame{extra}", flush=True)


def gen_only(K, ns):
    for f in range(F):
        gen_into(FR[f & 1], f, K, SIDE[:ns] if ns else None)


for rep in range(2):
    for K, ns in ((1, 0), (4, 0), (4, 2), (4, 3), (8, 3), (16, 3)):
        timed(f"generator alone, {K} chunk(s) on {ns or 1} stream(s)", lambda: gen_only(K, ns))
    timed("serial: generate, train, epilogue", lambda: pipeline("serial", 1, 0))
    timed("serial, generator 8 chunks on 3 streams", lambda: pipeline("serial", 8, 3))
    timed("phased: train | epilogue || generate (1 chunk)", lambda: pipeline("phased", 1, 0))
    timed("phased, generator 4 chunks on 3 streams", lambda: pipeline("phased", 4, 3))
    timed("phased, generator 8 chunks on 3 streams", lambda: pipeline("phased", 8, 3))
