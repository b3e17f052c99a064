"""Pins the CPU oracle (oracle/) against vectors captured from the reference itself.

Tolerances: the f32 oracle is the reference's arithmetic with a different summation
order, so it agrees with the reference's fp32 outputs to fp32 round-off amplified by
the steep softmin (|logit| ~ 1e3 => q to ~1e-4 of its max, see SURVEY 7 'Steep softmax');
the f64 oracle is the truth both sides are measured against.
"""
import numpy as np
import pytest

import oracle
from conftest import load_golden, relerr

G1 = ["G1_dp_step_64qam_pcs", "G1_dp_step_64qam", "G1_dp_step_16qam", "G1_dp_step_4qam", "G1_dp_step_64qam_nu0872", "G1_dp_step_64qam_nu1222"]   # the last two: config 5's heavy shaping (Eval_run_DP.py:24)


@pytest.mark.parametrize("name", G1)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dp_forward_loss_grads(name, dtype):
    g = load_golden(name)
    B, sps = int(g["B"]), int(g["sps"])
    x = g["rx"][:, :, :B * sps]
    r = oracle.dp_step_grads(x, g["W0"], g["h0"], g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), sps, dtype)
    assert relerr(r["out"], g["out0"]) < 2e-6
    assert np.max(np.abs(r["q"] - g["q0"])) < 2e-4            # q in [0,1]
    assert abs(r["loss"] - g["loss0"]) / abs(g["loss0"]) < 1e-5
    assert relerr(r["var_est"], g["var_est0"]) < 1e-5
    assert relerr(r["gh"], g["gh0"]) < 2e-5
    # SNR 28 dB with nu = 0.1222578: var = 1.5e-4 puts the logits at ~1e4, and the reference's OWN fp32 gradient sits 1.1e-4 of its
    # max away from the f64 truth (its fp32 restatement agrees with it to < 1e-4 like everywhere else)
    assert relerr(r["gW"], g["gW0"]) < (2e-4 if (dtype == np.float64 and name.endswith("nu1222")) else 1e-4)
    # the stand-alone entry points agree with the fused one
    q, out = oracle.dp_forward(x, g["W0"], g["amp_levels"], g["var"], float(g["nu_sc"]), sps, dtype)
    assert np.array_equal(q, r["q"]) and np.array_equal(out, r["out"])
    loss, ve = oracle.dp_loss(g["q0"], x, g["h0"], g["amp_levels"], g["P"], dtype)
    assert abs(loss - g["loss0"]) / abs(g["loss0"]) < 1e-5
    assert relerr(ve, g["var_est0"]) < 1e-5
    assert np.max(np.abs(oracle.dp_soft_dec(g["out0"], g["var"], g["amp_levels"], float(g["nu_sc"]), dtype) - g["q0"])) < 2e-4


@pytest.mark.parametrize("name", G1)
def test_dp_teacher_forced_adam(name):
    """Adam on the reference's own gradients reproduces the reference's updated taps (R5)."""
    g = load_golden(name)
    W, h = g["W0"].copy(), g["h0"].copy()
    mW, vW, mh, vh = np.zeros_like(W), np.zeros_like(W), np.zeros_like(h), np.zeros_like(h)
    for s in range(int(g["n_steps"])):
        oracle.adam(W, g[f"gW{s}"], mW, vW, s + 1, float(g["lr"]))
        oracle.adam(h, g[f"gh{s}"], mh, vh, s + 1, float(g["lr"]))
        assert np.max(np.abs(W - g[f"W{s + 1}"])) < 2e-7
        assert np.max(np.abs(h - g[f"h{s + 1}"])) < 2e-7
    assert relerr(mW, g["mW"]) < 1e-6 and relerr(vW, g["vW"]) < 1e-6
    assert relerr(mh, g["mh"]) < 1e-6 and relerr(vh, g["vh"]) < 1e-6


@pytest.mark.parametrize("name", G1)
def test_dp_three_steps_free(name):
    g = load_golden(name)
    B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
    st = oracle.DPState(M, np.float32, g["W0"], g["h0"])
    r = oracle.dp_train(st, g["rx"], 3, B, g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), float(g["lr"]), float(g["lr"]), sps)
    for s in range(3):
        assert abs(r["loss"][s] - g[f"loss{s}"]) / abs(g[f"loss{s}"]) < 2e-5
    assert np.max(np.abs(st.W - g["W3"])) < 2e-5
    assert np.max(np.abs(st.h - g["h3"])) < 2e-5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dp_freerun_vaele(dtype):
    """G2: 30 free-running VAE-LE steps from Dirac init (R6).  Taps <=1e-5 at step 20 (north_star)."""
    g = load_golden("G2_dp_freerun")
    B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
    st = oracle.DPState(M, dtype)
    lr = float(g["lr"])
    r = oracle.dp_train(st, g["rx"], 20, B, g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), lr, lr, sps, dtype=dtype)
    assert np.max(np.abs(r["loss"] - g["loss"][:20]) / np.abs(g["loss"][:20])) < 1e-5
    assert np.max(np.abs(st.W - g["W_after20"])) < 1e-5
    assert np.max(np.abs(st.h - g["h_after20"])) < 1e-5
    assert relerr(r["out"], g["out_const"][:, :, :20 * B]) < 3e-5
    assert np.max(np.abs(r["q"] - g["out_train"][:, :, :20 * B])) < 5e-4
    assert relerr(r["var_est"], g["var_est"][:, :20]) < 1e-5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dp_freerun_flex(dtype):
    """G3: 30 VAEflex window steps, stride 10, centre slice kept (R7)."""
    g = load_golden("G3_dp_flex_freerun")
    B, sps, M, fs, ns = int(g["B"]), int(g["sps"]), int(g["M_est"]), int(g["flex_step"]), int(g["n_steps"])
    st = oracle.DPState(M, dtype)
    lr = float(g["lr"])
    r = oracle.dp_train(st, g["rx"], ns, B, g["amp_levels"], g["P"], g["var"], float(g["nu_sc"]), lr, lr, sps,
                        stride=fs, keep_off=(B - fs) // 2, keep_len=fs, dtype=dtype)
    # strict over the first 10 steps, then the free-running trajectory drifts (Adam divides by sqrt(v):
    # the reference itself differs by 8e-5 between 1 and 4 CPU threads after 40 steps)
    assert np.max(np.abs(r["loss"][:10] - g["loss"][:10]) / np.abs(g["loss"][:10])) < 1e-5
    assert relerr(r["out"][:, :, :10 * fs], g["out_const"][:, :, :10 * fs]) < 1e-5
    assert np.max(np.abs(r["loss"] - g["loss"]) / np.abs(g["loss"])) < 1e-4
    assert relerr(r["out"], g["out_const"]) < 3e-4
    assert np.max(np.abs(r["q"] - g["out_train"])) < 2e-2
    assert relerr(r["var_est"], g["var_est"]) < 1e-4
    assert np.max(np.abs(st.W - g[f"W_after{ns}"])) < 3e-4
    assert np.max(np.abs(st.h - g[f"h_after{ns}"])) < 3e-4


AWGN = ["G4_awgn_16qam_cfg1", "G4_awgn_64qam_pcs_free10", "G4_awgn_4qam_small"]


@pytest.mark.parametrize("name", AWGN)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_awgn_step(name, dtype):
    g = load_golden(name)
    B, sps = int(g["B"]), int(g["sps"])
    x = g["rx"][:, :B * sps]
    r = oracle.awgn_step_grads(x, g["W0"], g["h0"], g["amp_levels"], g["P"], float(g["amp_mean"]), float(g["var"]), sps, dtype)
    assert relerr(r["out"], g["out0"]) < 2e-6
    assert np.max(np.abs(r["q"] - g["q0"])) < 5e-4
    assert abs(r["loss"] - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    assert relerr(r["gh"], g["gh0"]) < 2e-5
    assert relerr(r["gW"], g["gW0"]) < 1e-4
    assert abs(oracle.awgn_loss(g["q0"], x, g["h0"], g["amp_levels"], g["P"], dtype) - g["loss"][0]) / abs(g["loss"][0]) < 1e-5


@pytest.mark.parametrize("name", AWGN)
def test_awgn_freerun_amsgrad(name):
    g = load_golden(name)
    B, sps, M, ns = int(g["B"]), int(g["sps"]), int(g["M_est"]), int(g["n_steps"])
    st = oracle.AWGNState(M, np.float32, g["W0"], g["h0"])
    lr = float(g["lr"])
    loss = oracle.awgn_train(st, g["rx"], ns, B, g["amp_levels"], g["P"], float(g["amp_mean"]), float(g["var"]), lr, sps)
    # From the Dirac start the loss is exactly invariant to the scale of the single non-zero tap (the
    # demapper input is normalised by mean|y|, func_VAELE_MQAM_shaping.py:228), so that tap's true gradient
    # is 0 and the reference's first Adam step moves it by +-lr on rounding noise alone (golden gW0[0,0,M//2]
    # ~ -8e-6 against |gW| ~ 1e2).  Trajectories are therefore pinned up to that coin flip: strict only where
    # the start is not Dirac.
    dirac = np.count_nonzero(g["W0"]) == 1
    tol_l, tol_p = (2e-3, 2.5 * lr) if dirac else (2e-5, 2e-5)
    assert np.max(np.abs(loss - g["loss"]) / np.abs(g["loss"])) < tol_l
    assert np.max(np.abs(st.W - g[f"W{ns}"])) < tol_p
    assert np.max(np.abs(st.h - g[f"h{ns}"])) < tol_p
    if not dirac:
        assert relerr(st.vmaxW, g["vmaxW"]) < 1e-4 and relerr(st.vmaxh, g["vmaxh"]) < 1e-4


@pytest.mark.parametrize("name", AWGN)
def test_awgn_teacher_forced_amsgrad(name):
    """Adam(amsgrad=True) on the reference's own gradients reproduces its taps for the first 3 steps (R10)."""
    g = load_golden(name)
    W, h = g["W0"].copy(), g["h0"].copy()
    mW, vW, xW, mh, vh, xh = (np.zeros_like(a) for a in (W, W, W, h, h, h))
    for s in range(3):
        oracle.adam(W, g[f"gW{s}"], mW, vW, s + 1, float(g["lr"]), vmax=xW)
        oracle.adam(h, g[f"gh{s}"], mh, vh, s + 1, float(g["lr"]), vmax=xh)
        assert np.max(np.abs(W - g[f"W{s + 1}"])) < 2e-7
        assert np.max(np.abs(h - g[f"h{s + 1}"])) < 2e-7


def test_epilogue_converged_frame():
    """G5: shift search + both SER estimators on the last (converged) frame of a reference run (R12)."""
    g = load_golden("G5_dp_epilogue")
    r = oracle.dp_frame_epilogue(g["out_train"], g["out_const"], g["data"], g["amp_levels"], float(g["nu_sc"]), g["var"],
                                 batch_len=int(g["B"]))
    assert np.array_equal(r["shift_q"], g["shifts"][-1, 0]) and r["r_q"] == g["rs"][-1, 0]
    assert np.array_equal(r["shift_c"], g["shifts"][-1, 1]) and r["r_c"] == g["rs"][-1, 1]
    assert np.allclose(r["SER"], g["SER_valid"][:, -1], atol=1e-7)
    assert 0.005 < r["SER"].max() < 0.05      # the frame really is converged


@pytest.mark.parametrize("swap,delays", [(0, (2, -3)), (1, (2, 2)), (1, (-4, -4)), (0, (0, 0))])
def test_epilogue_polswap_and_shift(swap, delays):
    """Synthetic: (optionally) swap polarisations, delay, rotate by pi/2 -> the epilogue undoes all of it.

    With a polarisation swap the reference applies shift[0] of the PRE-swap pairing to the POST-swap row
    (func_VAELE_DP_MQAM_shaping.py:71-72), so only equal delays are undone there; restated as is."""
    g = load_golden("G5_dp_epilogue")
    amp, var, nu_sc = g["amp_levels"], g["var"], float(g["nu_sc"])
    rng = np.random.default_rng(3)
    N, n = 1000, amp.shape[0]
    lev = rng.integers(0, n, (2, 2, N))
    data = amp[lev].astype(np.float16)
    clean = amp[lev].astype(np.float32)
    rot = np.stack([-clean[:, 1], clean[:, 0]], 1)                       # +pi/2
    noisy = rot + 0.01 * rng.standard_normal(rot.shape).astype(np.float32)
    y = np.roll(noisy, swap, axis=0)
    y = np.stack([np.roll(y[0], delays[0], -1), np.roll(y[1], delays[1], -1)])
    q = oracle.dp_soft_dec(y, var, amp, nu_sc)
    r = oracle.dp_frame_epilogue(q, y, data, amp, nu_sc, var, batch_len=None)
    assert r["r_q"] == swap and r["r_c"] == swap
    assert tuple(r["shift_q"]) == delays and tuple(r["shift_c"]) == delays
    assert r["SER"].max() == 0.0


# ------------------------------------------------------------------ row f3: AWGN VAE-NN (G8)
G8 = ["G8_vaenn_64qam", "G8_vaenn_16qam_small", "G8_vaenn_4qam_k5"]


@pytest.mark.parametrize("name", G8)
def test_vaenn_forward_loss_grads(name):
    """Net.forward, loss_function and autograd's gradients of every parameter (fc1/fc2 weights and biases, h_est) at the
    Xavier-initialised start, teacher-forced on the captured minibatch."""
    g = load_golden(name)
    B, sps, k1, k2, M = int(g["B"]), int(g["sps"]), int(g["k1"]), int(g["k2"]), int(g["M_est"])
    n = len(g["amp_levels"])
    x = g["rx"][:, :B * sps]
    for dt, tq, tl, tg in ((np.float32, 2e-6, 2e-6, 2e-4), (np.float64, 2e-6, 2e-6, 2e-4)):
        q = oracle.nn_forward(x, g["theta0"], n, k1, k2, sps, dt)
        assert np.max(np.abs(q - g["q0"])) < tq
        h0 = oracle.nn_unpack(g["theta0"], n, k1, k2, M)[4]
        assert abs(oracle.nn_loss(g["q0"], x, h0, g["amp_levels"], dt) - g["loss"][0]) / abs(g["loss"][0]) < tl
        r = oracle.nn_step_grads(x, g["theta0"], g["amp_levels"], k1, k2, M, sps, dt)
        assert abs(r["loss"] - g["loss"][0]) / abs(g["loss"][0]) < tl
        gw1, gb1, gw2, gb2, gh = oracle.nn_unpack(r["g"], n, k1, k2, M)
        rw1, rb1, rw2, rb2, rh = oracle.nn_unpack(g["g0"], n, k1, k2, M)
        for a, b in ((gw1, rw1), (gb1, rb1), (gw2, rw2), (gb2, rb2), (gh, rh)):
            assert relerr(a, b) < tg


@pytest.mark.parametrize("name", G8)
def test_vaenn_free_run(name):
    """n_steps of the minibatch loop with Adam(amsgrad) on all 1650 parameters (64-QAM case): losses, parameters, moments."""
    g = load_golden(name)
    B, sps, k1, k2, M, ns = int(g["B"]), int(g["sps"]), int(g["k1"]), int(g["k2"]), int(g["M_est"]), int(g["n_steps"])
    st = oracle.NNState(g["theta0"], np.float32)
    loss = oracle.nn_train(st, g["rx"], ns, B, g["amp_levels"], k1, k2, M, float(g["lr"]), sps, np.float32)
    assert np.max(np.abs(loss[:3] - g["loss"][:3]) / np.abs(g["loss"][:3])) < 1e-5
    assert np.max(np.abs(loss - g["loss"]) / np.abs(g["loss"])) < 2e-3          # Adam amplifies rounding on near-zero gradients
    assert np.max(np.abs(st.theta - g[f"theta{ns}"])) < (5e-5 if ns <= 3 else 2 * ns * float(g["lr"]))
    st3 = oracle.NNState(g["theta0"], np.float32)
    oracle.nn_train(st3, g["rx"], 1, B, g["amp_levels"], k1, k2, M, float(g["lr"]), sps, np.float32)
    # one step: every parameter moves by ~lr in the direction of its gradient's sign; entries with a rounding-level gradient may flip
    ok = np.abs(g["g0"]) > 1e-4 * np.abs(g["g0"]).max()
    assert np.max(np.abs(st3.theta - g["theta1"])[ok]) < 2e-6
    assert np.max(np.abs(st3.theta - g["theta1"])) < 2.01 * float(g["lr"])


# ------------------------------------------------------------------ row f3, BatchNorm variant Net_BN (G11)
G11 = ["G11_vaennbn_64qam", "G11_vaennbn_16qam_small"]


@pytest.mark.parametrize("name", G11)
def test_vaennbn_step_and_running_statistics(name):
    """Net_BN: training-mode forward (batch statistics), all gradients incl. BatchNorm's gamma / beta, the running statistics after the
    step, the free run and the eval-mode forward on the running statistics."""
    g = load_golden(name)
    B, sps, k1, k2, M, ns = int(g["B"]), int(g["sps"]), int(g["k1"]), int(g["k2"]), int(g["M_est"]), int(g["n_steps"])
    n = len(g["amp_levels"])
    x = g["rx"][:, :B * sps]
    for dt in (np.float32, np.float64):
        r = oracle.nnbn_step_grads(x, g["theta0"], g["bn0"], g["amp_levels"], k1, k2, M, sps, dt)
        assert np.max(np.abs(r["q"] - g["q0"])) < 5e-6
        assert abs(r["loss"] - g["loss"][0]) / abs(g["loss"][0]) < 2e-6
        assert relerr(r["bn"], g["bn1"]) < 2e-6
        C_ = 2 * n
        o = np.cumsum([0, C_ * 2 * k1, C_, C_ * C_ * k2, C_, C_, C_, 2 * M])
        for a, b in zip(o[:-1], o[1:]):
            assert relerr(r["g"][a:b], g["g0"][a:b]) < 3e-4, (a, b)
    st = oracle.NNBNState(g["theta0"], n, np.float32)
    loss = oracle.nnbn_train(st, g["rx"], ns, B, g["amp_levels"], k1, k2, M, float(g["lr"]), sps, np.float32)
    assert np.max(np.abs(loss[:2] - g["loss"][:2]) / np.abs(g["loss"][:2])) < 2e-5
    assert np.max(np.abs(loss - g["loss"]) / np.abs(g["loss"])) < 2e-3
    assert relerr(st.bn, g[f"bn{ns}"]) < 1e-3
    assert np.max(np.abs(st.theta - g[f"theta{ns}"])) < 2 * ns * float(g["lr"])
    Ne = B * min(ns, 3)
    qe = oracle.nnbn_forward_eval(g["rx"][:, :Ne * sps], g[f"theta{ns}"], g[f"bn{ns}"], n, k1, k2, sps, np.float64)
    assert np.max(np.abs(qe - g["q_eval"])) < 5e-6


# ------------------------------------------------------------------ row f4: constant-modulus baselines + CPE (G12)
@pytest.mark.parametrize("tag,mode", [("cma", "CMA"), ("cmabatch", "CMAbatch"), ("cmaflex", "CMAflex")])
def test_cma_variants(tag, mode):
    """CMA / CMAbatch / CMAflex on one 700-symbol frame: outputs (incl. the wrapped first symbols), errors and updated taps."""
    g = load_golden("G12_cma")
    for dt, tol in ((np.float32, 2e-5), (np.float64, 5e-6)):
        h = g["h0"].astype(dt).copy()
        out, e = oracle.cma(g["rx"], h, float(g[f"lr_{tag}"]), int(g["sps"]), mode, int(g["batchlen"]), int(g["symb_step"]), 1.0, dt)
        assert np.isfinite(g[f"{tag}_out"]).all()
        assert relerr(out, g[f"{tag}_out"]) < tol and relerr(e, g[f"{tag}_e"]) < tol and relerr(h, g[f"{tag}_h"]) < tol
    assert np.abs(g[f"{tag}_h"] - g["h0"]).max() > 1e-3                       # the taps did move


def test_cpe_against_reference():
    g = load_golden("G12_cma")
    assert relerr(oracle.cpe(g["cpe_in"]), g["cpe_out"]) < 2e-6
    assert np.abs(g["cpe_out"] - g["cpe_in"]).max() > 0.3                      # a real de-rotation with unwrapped drift


@pytest.mark.parametrize("name", ["G14_cma_epilogue_64qam", "G14_cma_epilogue_16qam", "G14_cma_epilogue_64qam_pcs"])
def test_cma_frame_epilogue(name):
    """The CMA modules' two-stage epilogue incl. the in-place normalisation of the kept window (func_CMA_DP_MQAM_shaping.py:39-53,
    shared_funcs.py:242) on the last frame of a hand-driven reference loop; without the write-back the soft-demapper rows are far off at
    16- / 64-QAM (the CMA output settles at ~0.85 x the constellation's scale for R = 1)."""
    g = load_golden(name)
    amp, var, nu = g["amp_levels"], g["var"], float(g["nu_sc"])
    sd = lambda out, v, a, n: oracle.dp_soft_dec(out, v, a, n)
    r = oracle.cma_frame_epilogue(g["cma_out"], g["data"], amp, nu, var, sd)
    assert list(r["shift_c"]) == list(g["shifts"][-1, 0]) and r["r_c"] == g["rs"][-1, 0]
    assert list(r["shift_q"]) == list(g["shifts"][-1, 1]) and r["r_q"] == g["rs"][-1, 1]
    assert relerr(r["y"], g["out_const_after"]) < 2e-5
    assert np.max(np.abs(r["SER"] - g["SER"][:, -1])) < 1.5e-3, (r["SER"], g["SER"][:, -1])
    if name != "G14_cma_epilogue_64qam_pcs":          # what the test is for: the raw-scale flow is measurably different
        raw = oracle.cpe(g["cma_out"][:, :, 10:-10]).astype(np.float32)
        from oracle.epilogue import _align, find_shift, SER_IQflip
        ya = _align(raw, r["shift_c"], r["r_c"])
        q = oracle.dp_soft_dec(ya, var, amp, nu)
        s2, r2 = find_shift(q, g["data"][:, :, 10:-10], 21, amp)
        ms = int(np.max(np.abs(s2)))
        ser_raw = SER_IQflip(_align(q, s2, r2)[:, :, 11:-11 - ms], g["data"][:, :, 10:-10][:, :, 11:-11 - ms])
        assert np.min(ser_raw - g["SER"][2:, -1]) > 0.02, (ser_raw, g["SER"][2:, -1])
