"""GPU parity of the fused DP training kernel (through the C ABI) against the golden vectors captured from the
reference and against the CPU oracle on the same inputs.

Tolerances (fp32): per-minibatch ELBO <= 1e-5 relative (north_star); taps after teacher-forced / short free runs
<= 1e-5 absolute (taps are O(1)); q <= 2e-4 absolute (steep softmin); gradients <= 1e-4 of their max.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu

G1 = ["G1_dp_step_64qam_pcs", "G1_dp_step_64qam", "G1_dp_step_16qam", "G1_dp_step_4qam", "G1_dp_step_64qam_nu0872", "G1_dp_step_64qam_nu1222"]   # the last two: config 5's heavy shaping (Eval_run_DP.py:24)
DEV = "cuda:0"


def _engine(g, R=1, threads=0):
    from vae_equalizer_amd.engine import DPEngine
    return DPEngine(R, int(g["M_est"]), g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), DEV, int(g["sps"]), threads)


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("threads", [0, 64, 128, 256])
@pytest.mark.parametrize("name", G1)
def test_teacher_forced_step(name, threads):
    """One minibatch from a random (non-Dirac) state: q, y, ELBO, var_est, gradients, and the Adam update (R1-R5)."""
    g = load_golden(name)
    B, sps = int(g["B"]), int(g["sps"])
    eng = _engine(g, threads=threads)
    eng.set_state(g["W0"], g["h0"])
    rx = torch.from_numpy(g["rx"][None, :, :, :B * sps]).to(DEV)
    r = eng.train(rx, B, 1, float(g["lr"]), debug_grads=True)
    torch.cuda.synchronize()
    assert relerr(_np(r["y"])[0, 0], g["out0"]) < 2e-6
    assert np.max(np.abs(_np(r["q"])[0, 0] - g["q0"])) < 2e-4
    assert abs(_np(r["loss"])[0, 0, 0] - g["loss0"]) / abs(g["loss0"]) < 1e-5
    assert relerr(_np(r["var_est"])[0, 0, :, 0], g["var_est0"]) < 1e-5
    assert relerr(_np(r["gh"])[0], g["gh0"]) < 2e-5
    assert relerr(_np(r["gW"])[0], g["gW0"]) < 1e-4
    # truth check: at least as close to the f64 oracle as the reference's own fp32 result is (x3 slack)
    t = oracle.dp_step_grads(g["rx"][:, :, :B * sps], g["W0"], g["h0"], g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), sps, np.float64)
    assert relerr(_np(r["gW"])[0], t["gW"]) < max(3 * relerr(g["gW0"], t["gW"]), 2e-6)
    assert relerr(_np(r["gh"])[0], t["gh"]) < max(3 * relerr(g["gh0"], t["gh"]), 2e-6)
    assert abs(_np(r["loss"])[0, 0, 0] - t["loss"]) / abs(t["loss"]) < 1e-5
    # Adam: taps after the step; moments
    assert np.max(np.abs(_np(eng.W)[0] - g["W1"])) < 1e-5
    assert np.max(np.abs(_np(eng.h)[0] - g["h1"])) < 1e-5
    assert int(eng.step[0]) == 1


def test_step_debug_entry_point():
    """vaeq_dp_step_debug: one teacher-forced step that dumps dL/dW, dL/dh -- against the reference's autograd gradients (G1) and,
    bit for bit, against what vaeq_dp_train leaves in dbg_gW / dbg_gh."""
    import ctypes as C
    from vae_equalizer_amd import _native as nat
    g = load_golden(G1[0])
    B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
    rx = torch.from_numpy(g["rx"][None, None, :, :, :B * sps]).to(DEV).contiguous()
    eng = _engine(g)
    eng.set_state(g["W0"], g["h0"])
    ref = eng.train(rx, B, 1, float(g["lr"]), debug_grads=True, no_update=True)
    gW, gh = torch.zeros(1, 2, 4, M, device=DEV), torch.zeros(1, 2, 2, 2, M, device=DEV)
    lr = torch.full((1,), float(g["lr"]), device=DEV)
    loss = torch.zeros(1, 1, 1, device=DEV)
    a = nat.DPArgs(R=1, n_frames=1, steps=1, B=B, sps=sps, M=M, n_lev=eng.n_lev, stride_sym=B, keep_off=0, keep_len=B, S=B * sps, rx=nat.ptr(rx),
                   W=nat.ptr(eng.W), h=nat.ptr(eng.h), adam_mW=nat.ptr(eng.mW), adam_vW=nat.ptr(eng.vW), adam_mh=nat.ptr(eng.mh), adam_vh=nat.ptr(eng.vh),
                   step=nat.ptr(eng.step, torch.int32), amp=nat.ptr(eng.amp), P=nat.ptr(eng.P), var=nat.ptr(eng.var), nu_sc=nat.ptr(eng.nu_sc),
                   lr_W=nat.ptr(lr), lr_h=nat.ptr(lr), loss=nat.ptr(loss), threads=0, no_update=1)
    nat.check(nat.lib().vaeq_dp_step_debug(C.byref(a), nat.ptr(gW), nat.ptr(gh), nat.current_stream(torch.device(DEV))), "vaeq_dp_step_debug")
    torch.cuda.synchronize()
    assert torch.equal(gW, ref["gW"]) and torch.equal(gh, ref["gh"]) and torch.equal(loss, ref["loss"])
    assert relerr(_np(gh)[0], g["gh0"]) < 2e-5 and relerr(_np(gW)[0], g["gW0"]) < 1e-4
    assert int(eng.step[0]) == 0 and np.array_equal(_np(eng.W)[0], g["W0"].astype(np.float32))      # no_update: state untouched
    a.steps = 2
    assert nat.lib().vaeq_dp_step_debug(C.byref(a), nat.ptr(gW), nat.ptr(gh), nat.current_stream(torch.device(DEV))) != 0


@pytest.mark.parametrize("name", G1)
def test_three_steps_and_moments(name):
    g = load_golden(name)
    B = int(g["B"])
    eng = _engine(g)
    eng.set_state(g["W0"], g["h0"])
    rx = torch.from_numpy(g["rx"][None]).to(DEV)
    r = eng.train(rx, B, 3, float(g["lr"]))
    torch.cuda.synchronize()
    for s in range(3):
        assert abs(_np(r["loss"])[0, 0, s] - g[f"loss{s}"]) / abs(g[f"loss{s}"]) < 2e-5
    assert np.max(np.abs(_np(eng.W)[0] - g["W3"])) < 2e-5
    assert np.max(np.abs(_np(eng.h)[0] - g["h3"])) < 2e-5
    # Adam moments = running sums of the gradients: 1e-4 of their max, or -- where the reference's own fp32 gradients are noisier than that
    # (SNR 28 dB with nu = 0.1222578: logits ~1e4) -- at least as close to the f64 truth as the reference's moments are (x3 slack)
    st = oracle.DPState(int(g["M_est"]), np.float64, g["W0"], g["h0"])
    oracle.dp_train(st, g["rx"], 3, B, g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), float(g["lr"]), float(g["lr"]), int(g["sps"]), dtype=np.float64)
    for ours, key, truth in ((eng.mW, "mW", st.mW), (eng.vW, "vW", st.vW), (eng.mh, "mh", st.mh), (eng.vh, "vh", st.vh)):
        assert relerr(_np(ours)[0], g[key]) < 1e-4 or relerr(_np(ours)[0], truth) < 3 * relerr(g[key], truth), key
    assert int(eng.step[0]) == 3


def test_no_update_leaves_state():
    g = load_golden(G1[0])
    B, sps = int(g["B"]), int(g["sps"])
    eng = _engine(g)
    eng.set_state(g["W0"], g["h0"])
    rx = torch.from_numpy(g["rx"][None, :, :, :B * sps]).to(DEV)
    eng.train(rx, B, 1, float(g["lr"]), no_update=True)
    torch.cuda.synchronize()
    assert np.array_equal(_np(eng.W)[0], g["W0"]) and np.array_equal(_np(eng.h)[0], g["h0"])
    assert int(eng.step[0]) == 0 and float(eng.mW.abs().max()) == 0.0


@pytest.mark.parametrize("threads", [1, 64, 256])
def test_freerun_vaele_golden(threads):
    """G2: 20 free-running steps from the Dirac start: ELBO per minibatch and taps <= 1e-5 (north_star)."""
    g = load_golden("G2_dp_freerun")
    B = int(g["B"])
    eng = _engine(g, threads=threads)
    rx = torch.from_numpy(g["rx"][None]).to(DEV)
    r = eng.train(rx, B, 20, float(g["lr"]))
    torch.cuda.synchronize()
    loss = _np(r["loss"])[0, 0]
    assert np.max(np.abs(loss - g["loss"][:20]) / np.abs(g["loss"][:20])) < 1e-5
    assert np.max(np.abs(_np(eng.W)[0] - g["W_after20"])) < 1e-5
    assert np.max(np.abs(_np(eng.h)[0] - g["h_after20"])) < 1e-5
    assert relerr(_np(r["y"])[0, 0], g["out_const"][:, :, :20 * B]) < 3e-5
    assert np.max(np.abs(_np(r["q"])[0, 0] - g["out_train"][:, :, :20 * B])) < 5e-4
    assert relerr(_np(r["var_est"])[0, 0], g["var_est"][:, :20]) < 1e-5


@pytest.mark.parametrize("threads", [1, 256])
def test_freerun_flex_golden(threads):
    """G3: VAEflex windows (stride 10, centre slice kept), 30 steps (threads=1: wave-per-run kernel, scalar-store variant)."""
    g = load_golden("G3_dp_flex_freerun")
    B, fs, ns = int(g["B"]), int(g["flex_step"]), int(g["n_steps"])
    eng = _engine(g, threads=threads)
    rx = torch.from_numpy(g["rx"][None]).to(DEV)
    r = eng.train(rx, B, ns, float(g["lr"]), stride=fs, keep_off=(B - fs) // 2, keep_len=fs)
    torch.cuda.synchronize()
    loss = _np(r["loss"])[0, 0]
    assert np.max(np.abs(loss[:10] - g["loss"][:10]) / np.abs(g["loss"][:10])) < 1e-5
    assert relerr(_np(r["y"])[0, 0][:, :, :10 * fs], g["out_const"][:, :, :10 * fs]) < 1e-5
    # 30 free steps against the reference, measured on the GPU with the wave kernel (printed below): loss 5.6e-6 relative, y 9.7e-5 of its max, taps
    # W 2.5e-5 / h 1.6e-5 absolute; the reference's own noise at this horizon is ~2e-5 on the taps (1 vs 4 CPU threads).
    # Bounds = 3 x the measured values.
    dl, dy = np.max(np.abs(loss - g["loss"]) / np.abs(g["loss"])), relerr(_np(r["y"])[0, 0], g["out_const"])
    dW, dh = np.max(np.abs(_np(eng.W)[0] - g[f"W_after{ns}"])), np.max(np.abs(_np(eng.h)[0] - g[f"h_after{ns}"]))
    print(f"flex free run, {ns} steps, threads={threads}: loss rel {dl:.2e}  y rel {dy:.2e}  W abs {dW:.2e}  h abs {dh:.2e}")
    assert dl < 2e-5 and dy < 3e-4 and dW < 8e-5 and dh < 8e-5, (dl, dy, dW, dh)


def test_batch_of_runs_matches_oracle():
    """R=37 runs with different seeds / SNR / shaping / lr in one launch == 37 oracle runs (ragged sizes: B=37, M=9)."""
    from vae_equalizer_amd.engine import DPEngine
    rng = np.random.default_rng(5)
    R, B, M, n, sps, steps = 37, 37, 9, 4, 2, 6
    amp = np.array([-3, -1, 1, 3], np.float32) / np.sqrt(10).astype(np.float32)
    P = rng.dirichlet(np.ones(n) * 5, R).astype(np.float32)
    var = rng.uniform(0.002, 0.02, (R, 2)).astype(np.float32)
    nu_sc = rng.uniform(0, 1, R).astype(np.float32)
    lr = rng.uniform(1e-3, 4e-3, R).astype(np.float32)
    rx = (0.4 * rng.standard_normal((R, 2, 2, steps * B * sps))).astype(np.float32)
    eng = DPEngine(R, M, amp, P, var, nu_sc, DEV, sps)
    r = eng.train(torch.from_numpy(rx).to(DEV), B, steps, lr)
    torch.cuda.synchronize()
    for i in range(R):
        st = oracle.DPState(M, np.float32)
        o = oracle.dp_train(st, rx[i], steps, B, amp, P[i], var[i], float(nu_sc[i]), float(lr[i]), float(lr[i]), sps)
        assert np.max(np.abs(_np(r["loss"])[i, 0] - o["loss"]) / np.abs(o["loss"])) < 2e-5, i
        assert np.max(np.abs(_np(eng.W)[i] - st.W)) < 2e-5, i
        assert np.max(np.abs(_np(eng.h)[i] - st.h)) < 2e-5, i
        assert relerr(_np(r["y"])[i, 0], o["out"]) < 2e-5, i


def test_multi_frame_call_equals_frame_by_frame():
    """n_frames=3 in one launch == three launches with state carried by the caller (bitwise)."""
    g = load_golden("G2_dp_freerun")
    B, sps = int(g["B"]), int(g["sps"])
    rx = torch.from_numpy(g["rx"][:, :, :3 * 5 * B * sps]).to(DEV)
    frames = rx.reshape(2, 2, 3, 5 * B * sps).permute(2, 0, 1, 3).contiguous()   # [F,2,2,S]
    e1, e2 = _engine(g), _engine(g)
    r1 = e1.train(frames[None], B, 5, float(g["lr"]))
    l2 = [e2.train(frames[f][None, None], B, 5, float(g["lr"]))["loss"] for f in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(r1["loss"][0], torch.cat([x[0] for x in l2]))
    assert torch.equal(e1.W, e2.W) and torch.equal(e1.h, e2.h) and int(e1.step[0]) == 15


@pytest.mark.parametrize("B,threads,flex", [(100, 0, False), (200, 0, False), (100, 64, False), (100, 0, True), (50, 0, False)])
def test_frames_per_launch_do_not_change_the_results(B, threads, flex):
    """How many frames a launch holds is a scheduling choice (dp_runs groups the frames of small sweeps): 6 frames x 40 steps as ONE launch, as
    launches of 4 + 2 and frame by frame leave the same taps, Adam state and outputs, bit for bit -- beta^t of the Adam bias corrections restarts
    from pow() at every frame head exactly as at a launch (wave kernel, two-wave kernel, generic kernel, VAEflex windows; runs that are 40 and
    4000 steps old, where 1 - 0.999^t still moves)."""
    from vae_equalizer_amd import shared_funcs as sfun
    from vae_equalizer_amd.engine import DPEngine
    t = sfun.qam_tables("64-QAM", 0.0270955)
    R, F, steps = 5, 6, 40
    stride, klen, k0 = (10, 10, (B - 10) // 2) if flex else (B, B, 0)
    S = 2 * ((steps - 1) * stride + B)
    g = torch.Generator(device="cpu").manual_seed(B + threads)
    rx = (0.35 * torch.randn(R, F, 2, 2, S, generator=g)).to(DEV)
    var = t["pow_mean"] / 10 ** 2.3 / 2

    def run(groups):
        eng = DPEngine(R, 25, t["amps"], t["P"], [var, var], t["nu_sc"], DEV, 2, threads)
        eng.step[1:] = 4000                                                       # bias corrections of old runs: 0.999^t = 0.018 ... 0.015
        eng.train(rx[:, 0].contiguous(), B, steps, 2.5e-3, stride=stride, keep_off=k0, keep_len=klen, want_q=False, want_compact=True)   # warm state
        outs = []
        for lo, hi in groups:
            o = eng.train(rx[:, lo:hi].contiguous(), B, steps, 2.5e-3, stride=stride, keep_off=k0, keep_len=klen, want_q=False, want_compact=True)
            outs.append({k: o[k] for k in ("loss", "var_est", "eq", "dec", "y")})
        torch.cuda.synchronize()
        cat = {k: torch.cat([o[k] for o in outs], dim=1) for k in outs[0]}
        return eng, cat
    ea, a = run([(0, 6)])
    for groups in ([(0, 4), (4, 6)], [(f, f + 1) for f in range(6)]):
        eb, b = run(groups)
        for k in a:
            assert torch.equal(a[k], b[k]), (k, groups)
        for k in ("W", "h", "mW", "vW", "mh", "vh", "step"):
            assert torch.equal(getattr(ea, k), getattr(eb, k)), (k, groups)
    assert torch.isfinite(a["loss"]).all() and int(ea.step[0]) == 7 * steps and int(ea.step[1]) == 4000 + 7 * steps


def test_deterministic_bitwise():
    g = load_golden("G2_dp_freerun")
    rx = torch.from_numpy(g["rx"][None]).to(DEV)
    outs = []
    for _ in range(2):
        eng = _engine(g)
        r = eng.train(rx, int(g["B"]), 30, float(g["lr"]))
        torch.cuda.synchronize()
        outs.append((r["loss"].clone(), eng.W.clone(), r["q"].clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_soft_demap_and_forward_entry_points():
    from vae_equalizer_amd.engine import dp_forward, soft_demap
    g = load_golden(G1[0])
    B, sps = int(g["B"]), int(g["sps"])
    q = soft_demap(torch.from_numpy(g["out0"]).to(DEV), g["amp_levels"], g["var"], float(g["nu_sc"]))
    assert np.max(np.abs(_np(q) - g["q0"])) < 2e-4
    q2, y2 = dp_forward(torch.from_numpy(g["rx"][:, :, :B * sps]).to(DEV), torch.from_numpy(g["W0"]).to(DEV), g["amp_levels"],
                        g["var"], float(g["nu_sc"]), sps)
    assert relerr(_np(y2), g["out0"]) < 2e-6
    assert np.max(np.abs(_np(q2) - g["q0"])) < 2e-4


def test_error_codes():
    from vae_equalizer_amd import _native as nat
    from vae_equalizer_amd.engine import DPEngine
    g = load_golden(G1[0])
    eng = _engine(g)
    rx = torch.zeros(1, 1, 2, 2, 100, device=DEV)
    with pytest.raises(nat.VaeqError):          # window (B=100 symbols -> 200 samples) past the row
        eng.train(rx, 100, 1, 1e-3)
    with pytest.raises(ValueError):
        DPEngine(1, 24, g["amp_levels"], g["P"], g["var"], 0.0, DEV)
    with pytest.raises(nat.VaeqError):          # CPU tensors are refused: there is no CPU path
        eng.train(torch.zeros(1, 1, 2, 2, 400), 100, 1, 1e-3)


@pytest.mark.parametrize("B,M,n", [(20, 25, 8), (14, 25, 4), (24, 25, 8), (10, 9, 8), (8, 13, 2), (16, 31, 4), (100, 25, 8), (64, 25, 4), (128, 13, 2), (50, 9, 8), (26, 25, 8), (100, 31, 8), (80, 17, 4), (60, 21, 2),
                                   (128, 25, 8), (64, 25, 8), (128, 25, 2), (50, 25, 8), (98, 25, 4), (126, 25, 8), (254, 25, 8), (300, 25, 8), (770, 25, 4), (130, 25, 8), (200, 25, 8), (256, 13, 2), (258, 31, 4), (300, 9, 8), (384, 21, 2), (512, 25, 8), (154, 17, 4), (600, 13, 4), (1000, 31, 2), (1024, 25, 8),
                                   # odd minibatch lengths (the last lane's symbol pair is half empty; rows start 8-byte aligned): every size class and tap count
                                   (99, 25, 8), (101, 25, 8), (127, 25, 4), (25, 25, 8), (15, 9, 8), (13, 25, 2), (77, 17, 4), (33, 31, 4), (129, 25, 8), (255, 13, 2), (257, 31, 4),
                                   (301, 9, 8), (511, 25, 8), (513, 21, 2), (999, 17, 4), (1023, 25, 8)])
def test_wave_kernel_equals_generic_kernel(B, M, n):
    """The wave-per-run fast path (threads=1; one wavefront per run up to B = 128, two up to 256, four up to 512, eight up to 1024; M = 25: B = 100 /
    128 / 200 / 400 baked, every other B -- even or odd -- on the fixed LDS layout of its size class, like all B of the other tap counts) and the generic
    kernel (threads=256) agree on ragged shapes, 5 free steps, R=9."""
    from vae_equalizer_amd.engine import DPEngine
    rng = np.random.default_rng(B + M)
    R, sps, steps = 9, 2, 5
    lev = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = lev / np.sqrt(np.mean(lev ** 2) * 2).astype(np.float32)
    P = rng.dirichlet(np.ones(n) * 5, R).astype(np.float32)
    var = rng.uniform(0.002, 0.02, (R, 2)).astype(np.float32)
    nu_sc = rng.uniform(0, 1, R).astype(np.float32)
    lr = rng.uniform(1e-3, 4e-3, R).astype(np.float32)
    rx = torch.from_numpy((0.4 * rng.standard_normal((R, 2, 2, steps * B * sps))).astype(np.float32)).to(DEV)
    outs = []
    for th in (1, 256):
        eng = DPEngine(R, M, amp, P, var, nu_sc, DEV, sps, th)
        eng.set_state(eng.W + 0.03 * torch.randn(eng.W.shape, generator=torch.Generator().manual_seed(1)).to(DEV),
                      eng.h + 0.03 * torch.randn(eng.h.shape, generator=torch.Generator().manual_seed(2)).to(DEV))
        r = eng.train(rx, B, steps, lr, debug_grads=True)
        torch.cuda.synchronize()
        outs.append((r, eng))
    (ra, ea), (rb, eb) = outs
    assert relerr(_np(ra["loss"]), _np(rb["loss"])) < 2e-6
    assert relerr(_np(ra["y"]), _np(rb["y"])) < 1e-5
    assert np.max(np.abs(_np(ra["q"]) - _np(rb["q"]))) < 2e-4
    assert relerr(_np(ra["var_est"]), _np(rb["var_est"])) < 1e-5
    assert relerr(_np(ra["gW"]), _np(rb["gW"])) < 1e-3 and relerr(_np(ra["gh"]), _np(rb["gh"])) < 1e-3     # gradients of the 5th free step
    assert np.max(np.abs(_np(ea.W) - _np(eb.W))) < 2e-5 and np.max(np.abs(_np(ea.h) - _np(eb.h))) < 2e-5
    assert relerr(_np(ea.mW), _np(eb.mW)) < 1e-3 and relerr(_np(ea.vh), _np(eb.vh)) < 1e-3
    assert torch.equal(ea.step, eb.step)


@pytest.mark.parametrize("B,k0,klen", [(200, 95, 11), (200, 90, 20), (300, 145, 10), (140, 65, 11), (700, 345, 11), (128, 59, 11), (64, 26, 12), (99, 44, 10), (255, 122, 11), (513, 250, 12)])
def test_multiwave_flex_windows_equal_generic_kernel(B, k0, klen):
    """VAEflex windows (stride 10, centre slice kept; odd offsets take the scalar-store variant) on the two- / four-wave kernels (and the baked B = 64 / 128 single-wave shapes),
    with the compact epilogue outputs, against the generic kernel: 12 free steps, R = 5."""
    from vae_equalizer_amd.engine import DPEngine
    rng = np.random.default_rng(B + k0)
    R, sps, M, n, stride, steps = 5, 2, 25, 8, 10, 12
    lev = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = lev / np.sqrt(np.mean(lev ** 2) * 2).astype(np.float32)
    P = rng.dirichlet(np.ones(n) * 5, R).astype(np.float32)
    var = rng.uniform(0.002, 0.02, (R, 2)).astype(np.float32)
    rx = torch.from_numpy((0.4 * rng.standard_normal((R, 2, 2, ((steps - 1) * stride + B) * sps + 8))).astype(np.float32)).to(DEV)
    outs = []
    for th in (1, 256):
        eng = DPEngine(R, M, amp, P, var, rng.uniform(0, 1, R).astype(np.float32) * 0 + 0.3, DEV, sps, th)
        r = eng.train(rx, B, steps, 2e-3, stride=stride, keep_off=k0, keep_len=klen, want_compact=True)
        torch.cuda.synchronize()
        outs.append((r, eng))
    (ra, ea), (rb, eb) = outs
    assert ra["y"].shape == rb["y"].shape and ra["y"].shape[-1] == steps * klen
    assert relerr(_np(ra["loss"]), _np(rb["loss"])) < 2e-6 and relerr(_np(ra["y"]), _np(rb["y"])) < 1e-5
    assert np.max(np.abs(_np(ra["q"]) - _np(rb["q"]))) < 2e-4 and relerr(_np(ra["eq"]), _np(rb["eq"])) < 1e-4
    assert np.mean(_np(ra["dec"]) != _np(rb["dec"])) < 2e-3                     # argmax ties at decision boundaries only
    assert np.max(np.abs(_np(ea.W) - _np(eb.W))) < 2e-5 and np.max(np.abs(_np(ea.h) - _np(eb.h))) < 2e-5


@pytest.mark.parametrize("B,M,n", [(200, 25, 4), (600, 13, 4), (400, 25, 8), (256, 31, 8), (130, 25, 8), (1000, 25, 8), (100, 25, 8), (128, 21, 2),
                                   (128, 25, 8), (64, 25, 4)])
def test_multiwave_against_oracle_frames_and_determinism(B, M, n):
    """One / two / four / eight wavefronts per run (B <= 128 / 256 / 512 / 1024), 4- to 64-QAM: each run == the CPU oracle; two frames in one
    launch == two launches (bitwise); a repeat is bitwise identical (fixed-order cross-wave sums); no_update leaves taps, moments and the step
    counter alone."""
    from vae_equalizer_amd.engine import DPEngine
    rng = np.random.default_rng(B)
    R, sps, steps = 4, 2, 3
    lev = np.arange(-(n - 1), n, 2).astype(np.float64)
    amp = (lev / np.sqrt(2 * np.mean(lev ** 2))).astype(np.float32)
    P = rng.dirichlet(np.ones(n) * 5, R).astype(np.float32)
    var = rng.uniform(0.002, 0.02, (R, 2)).astype(np.float32)
    nu_sc = rng.uniform(0, 1, R).astype(np.float32)
    lr = rng.uniform(1e-3, 4e-3, R).astype(np.float32)
    rx = (0.4 * rng.standard_normal((R, 2, 2, 2, steps * B * sps))).astype(np.float32)          # [R, F=2, 2, 2, S]
    rxd = torch.from_numpy(rx).to(DEV)
    mk = lambda: DPEngine(R, M, amp, P, var, nu_sc, DEV, sps, 1)
    e1, e2, e3 = mk(), mk(), mk()
    r1 = e1.train(rxd, B, steps, lr)
    r2 = [e2.train(rxd[:, f:f + 1].contiguous(), B, steps, lr) for f in range(2)]
    r3 = e3.train(rxd, B, steps, lr)
    torch.cuda.synchronize()
    assert torch.equal(r1["loss"], torch.cat([x["loss"] for x in r2], 1)) and torch.equal(e1.W, e2.W) and torch.equal(e1.h, e2.h)
    assert torch.equal(r1["loss"], r3["loss"]) and torch.equal(r1["q"], r3["q"]) and torch.equal(e1.W, e3.W) and torch.equal(e1.mh, e3.mh)
    for i in range(R):
        st = oracle.DPState(M, np.float32)
        lo = [oracle.dp_train(st, rx[i, f], steps, B, amp, P[i], var[i], float(nu_sc[i]), float(lr[i]), float(lr[i]), sps) for f in range(2)]
        assert np.max(np.abs(_np(r1["loss"])[i, 1] - lo[1]["loss"]) / np.abs(lo[1]["loss"])) < 2e-5, i
        assert np.max(np.abs(_np(e1.W)[i] - st.W)) < 2e-5 and np.max(np.abs(_np(e1.h)[i] - st.h)) < 2e-5, i
        assert relerr(_np(r1["y"])[i, 1], lo[1]["out"]) < 2e-5, i
    W0, h0, m0, st0 = e1.W.clone(), e1.h.clone(), e1.mW.clone(), e1.step.clone()
    e1.train(rxd, B, steps, lr, no_update=True)
    torch.cuda.synchronize()
    assert torch.equal(e1.W, W0) and torch.equal(e1.h, h0) and torch.equal(e1.mW, m0) and torch.equal(e1.step, st0)


def test_wave_kernel_refused_for_unsupported_shape():
    from vae_equalizer_amd import _native as nat
    from vae_equalizer_amd.engine import DPEngine
    g = load_golden("G1_dp_step_4qam")          # (B = 37: odd minibatch lengths run on the wave kernel since round 3)
    eng = DPEngine(1, 11, g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), DEV, 2, threads=1)       # a tap count without an instantiation
    with pytest.raises(nat.VaeqError):
        eng.train(torch.from_numpy(g["rx"][None]).to(DEV), int(g["B"]), 1, 1e-3)
    eng = DPEngine(1, int(g["M_est"]), g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), DEV, 3, threads=1)   # sps = 3
    with pytest.raises(nat.VaeqError):
        eng.train(torch.zeros(1, 1, 2, 2, 3 * int(g["B"]), device=DEV), int(g["B"]), 1, 1e-3)


# ------------------------------------------------------------------ compact epilogue inputs (eq_out / dec_out)
@pytest.mark.parametrize("threads", [1, 256])
@pytest.mark.parametrize("flex", [False, True])
def test_compact_outputs_equal_what_the_epilogue_derives_from_q(threads, flex):
    """eq_out = E_q[x_I] and dec_out = argmax(q) written by the training kernel are bit-identical to the values the epilogue computes
    from the materialised q of the same call, so vaeq_dp_epilogue_compact returns exactly vaeq_dp_epilogue's SER / shifts -- with
    or without q being written at all."""
    from vae_equalizer_amd import channel as ch, shared_funcs as sfun
    from vae_equalizer_amd.engine import DPEngine, dp_epilogue, dp_epilogue_compact
    R, B, M, sps, N = 5, 100, 25, 2, 3000
    t = sfun.qam_tables("64-QAM", 0.0270955)
    h_ch = sfun.upsampled_channel("h0", sps)
    var = t["pow_mean"] / 10 ** 2.3 / 2
    rx, data = ch.generate_batch_hip(R, N, t["amps"], t["P"], 23.0, h_ch, 90e9, sps, -26e-24, 0.1e-12 * np.sqrt(1000),
                                     np.array([0.0314, 0.0314], np.complex64), np.linspace(0.1, 1.2, R), DEV, 5, 0)
    if flex:
        stride, klen, k0 = 10, 10, (B - 10) // 2
        steps = (N - B) // stride
        data = data[:, :, :, B // 2:steps * stride + B // 2]
    else:
        stride, klen, k0, steps = B, B, 0, N // B
    outs = []
    for want_q in (True, False):
        eng = DPEngine(R, M, t["amps"], t["P"], [var, var], t["nu_sc"], DEV, sps, threads)
        outs.append(eng.train(rx, B, steps, 2.5e-3, stride=stride, keep_off=k0, keep_len=klen, want_q=want_q, want_compact=True))
    torch.cuda.synchronize()
    a, b = outs
    assert b["q"] is None and torch.equal(a["eq"], b["eq"]) and torch.equal(a["dec"], b["dec"]) and torch.equal(a["y"], b["y"])
    q = a["q"][:, 0]
    n = q.shape[2] // 2
    amp = torch.tensor(t["amps"], dtype=torch.float32, device=DEV)
    assert torch.equal(a["dec"][:, 0].long(), torch.stack([q[:, 0, :n].argmax(1), q[:, 0, n:].argmax(1), q[:, 1, :n].argmax(1),
                                                           q[:, 1, n:].argmax(1)], 1).reshape(R, 2, 2, -1))
    assert relerr(a["eq"][:, 0].cpu().numpy(), torch.einsum("i,rpin->rpn", amp, q[:, :, :n]).cpu().numpy()) < 1e-6
    nu = torch.full((R,), float(t["nu_sc"]), device=DEV)
    varr = torch.full((R, 2), var, device=DEV)
    bl = None if flex else B
    ref = dp_epilogue(q, a["y"][:, 0], data, amp, nu, varr, bl)
    got = dp_epilogue_compact(b["eq"][:, 0], b["dec"][:, 0], b["y"][:, 0], data, amp, nu, varr, bl)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
