#!/usr/bin/env python3
"""Can the drop-in pipeline at 8192 runs hide its channel simulator and epilogue BESIDE the training kernel when that kernel leaves room on the chip?

The training kernel holds 2 x 254 of the 512 vector registers of every SIMD for a whole 2048-run round, so side streams only time-slice the chip
(tools/probe_pipeline_overlap.py).  Here a frame's training call is issued as CHUNKS of runs (vaeq_dp_train on contiguous run ranges of the same
state / frame / output tensors): at 1024 runs per launch one wavefront sits on each SIMD (12.8 us per step instead of 20.5 us for two sharing it)
and half of the register file plus 97 KB of LDS per CU stay free for the generator passes of frame f + 1 and the epilogue of frame f - 1 on side streams.
Modes: serial (one stream, full launches), overlap (three streams, full launches), chunked-serial (one stream, chunks), chunked-overlap (three
streams, chunks on a high-priority stream).  Identical results are asserted (SER checksum).  GPU box only.

usage: probe_pipeline_chunked.py [R=8192] [F=8] [chunk=1024,2048]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_equalizer_amd import channel as ch, shared_funcs as sfun, _native as nat
from vae_equalizer_amd.engine import DPEngine, dp_epilogue_compact

dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
CHUNKS = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1024, 2048]
CFG = bench.CFG
t = sfun.qam_tables(CFG["mod"], CFG["nu"]); h_ch = sfun.upsampled_channel(CFG["channel"], 2)
var = t["pow_mean"] / 10 ** 2.3 / 2
amp = torch.tensor(t["amps"], dtype=torch.float32, device=dev)
nu = torch.zeros(R, device=dev); varr = torch.full((R, 2), var, device=dev)
B, STEPS, NO = 100, 100, 10000


def gen(f):
    return ch.generate_batch_hip(R, 10000, t["amps"], t["P"], 23.0, h_ch, 90e9, 2, CFG["tau_cd"], CFG["tau_pmd"], CFG["phiIQ"], 0.3 + 0.06 * np.pi * f, dev, 1, f)


def new_out():
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    return {"y": e(R, 1, 2, 2, NO), "eq": e(R, 1, 2, NO), "dec": torch.empty(R, 1, 2, 2, NO, dtype=torch.int8, device=dev),
            "loss": e(R, 1, STEPS), "var_est": e(R, 1, 2, STEPS)}


def train_range(eng, rx, out, lr, r0, r1):
    """vaeq_dp_train on runs [r0, r1) of the engine's state: every per-run array is contiguous along the run axis, so a range is a pointer offset."""
    s = slice(r0, r1)
    rx5 = rx if rx.dim() == 5 else rx.unsqueeze(1)
    a = nat.DPArgs(R=r1 - r0, n_frames=1, steps=STEPS, B=B, sps=2, M=eng.M, n_lev=eng.n_lev, stride_sym=B, keep_off=0, keep_len=B, S=rx5.shape[-1],
                   rx=nat.ptr(rx5[s]), W=nat.ptr(eng.W[s]), h=nat.ptr(eng.h[s]), adam_mW=nat.ptr(eng.mW[s]), adam_vW=nat.ptr(eng.vW[s]),
                   adam_mh=nat.ptr(eng.mh[s]), adam_vh=nat.ptr(eng.vh[s]), step=nat.ptr(eng.step[s], torch.int32), amp=nat.ptr(eng.amp),
                   P=nat.ptr(eng.P[s]), var=nat.ptr(eng.var[s]), nu_sc=nat.ptr(eng.nu_sc[s]), lr_W=nat.ptr(lr[s]), lr_h=nat.ptr(lr[s]),
                   q_out=None, y_out=nat.ptr(out["y"][s]), loss=nat.ptr(out["loss"][s]), var_est=nat.ptr(out["var_est"][s]),
                   eq_out=nat.ptr(out["eq"][s]), dec_out=nat.ptr(out["dec"][s], torch.int8), dbg_gW=None, dbg_gh=None, threads=0, no_update=0)
    nat.check(nat.lib().vaeq_dp_train(C.byref(a), nat.current_stream(dev)), "vaeq_dp_train")


def train(eng, rx, out, lr, chunk):
    for r0 in range(0, R, chunk):
        train_range(eng, rx, out, lr, r0, min(R, r0 + chunk))


def epi(out, data):
    return dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu, varr, 100)["SER"]


def serial(chunk):
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
    lr = torch.full((R,), 2.5e-3, device=dev)
    sers = []
    for f in range(F):
        rx, data = gen(f)
        out = new_out()
        train(eng, rx, out, lr, chunk)
        sers.append(epi(out, data))
    return torch.stack(sers)


def overlapped(chunk, prio):
    eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], dev, 2)
    lr = torch.full((R,), 2.5e-3, device=dev)
    cur = torch.cuda.current_stream(dev)
    main = HI if prio else cur
    gs, es = GS, ES
    sers, keep = [], []
    main.wait_stream(cur); gs.wait_stream(cur); es.wait_stream(cur)
    with torch.cuda.stream(gs):
        nxt = gen(0); g_done = torch.cuda.Event(); g_done.record(gs)
    for f in range(F):
        main.wait_event(g_done)
        rx, data = nxt
        with torch.cuda.stream(main):
            out = new_out()
            train(eng, rx, out, lr, chunk)
            t_done = torch.cuda.Event(); t_done.record(main)
        if f + 1 < F:
            with torch.cuda.stream(gs):
                nxt = gen(f + 1); g_done = torch.cuda.Event(); g_done.record(gs)
        es.wait_event(t_done)
        with torch.cuda.stream(es):
            sers.append(epi(out, data))
        keep.append((out, rx, data))                           # everything stays alive: no buffer is recycled while a side stream may still read it
    cur.wait_stream(main); cur.wait_stream(es); cur.wait_stream(gs)
    torch.cuda.synchronize()
    return torch.stack(sers)


HI, GS, ES = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev), torch.cuda.Stream(dev)   # created once: the caching allocator pools memory per stream


def timed(name, fn):
    fn()                                                        # every mode once untimed: its streams' pools are filled
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = fn()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{name:34s} R={R} {F} frames: {1e3 * el / F:7.2f} ms per frame  {R * 10000 * F / el / 1e9:.3f} G DP symbols/s   SER checksum {float(S.double().sum()):.6f}", flush=True)
    return S


serial(R)                                                       # warm-up: code objects, generator tables, allocator
ref = None
for rep in range(2):
    S = timed("serial, full launches", lambda: serial(R)); ref = S if ref is None else ref
    assert torch.equal(S, ref)
    S = timed("three streams, full launches", lambda: overlapped(R, False)); assert torch.equal(S, ref)
    for c in CHUNKS:
        S = timed(f"serial, {c} runs per launch", lambda: serial(c)); assert torch.equal(S, ref)
        S = timed(f"three streams, {c} runs per launch", lambda: overlapped(c, False)); print("   identical:", torch.equal(S, ref))
        S = timed(f"three streams + priority, {c} per launch", lambda: overlapped(c, True)); print("   identical:", torch.equal(S, ref))
