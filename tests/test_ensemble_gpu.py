"""Run-level parity as a STATISTICAL statement: the HIP path against the f32 C oracle on the same ensemble of seeded frame sets.

A free-running blind equaliser is chaotic in fp32 (the reference against itself at 1 vs 4 CPU threads: 2.6e-2 in W after 100 steps, SURVEY
section 7; GPU vs oracle: the tap deviation doubles every 3-4 steps, profiles/r02_final/parity_vs_free_steps.txt), so WHEN one run escapes the
initial plateau cannot be pinned by a single seed -- rounding-level changes move it by tens of frames.  What can be pinned is the
distribution: K runs on K seeded frame sets go through the product path (device generator -> vaeq_dp_train -> vaeq_dp_epilogue_compact,
what processing() runs) and, frame by frame on the SAME samples, through the oracle (oracle.dp_train_batch_f32 + the numpy epilogue), from the
same Dirac start (func_VAELE_DP_MQAM_shaping.py:26-31,43-89).  Asserted:
  (a) before chaos sets in (frame 0) the two agree run by run;
  (b) the escape-frame distributions agree (two-sample Kolmogorov-Smirnov, medians);
  (c) converged SER and noise estimate agree within 3 sigma of the Monte-Carlo error of the ensemble means;
  (d) the reference's own captured run (tests/golden, one seed on the host generator) lies inside the ensemble's range.
The single-seed tests of test_processing_gpu.py keep the early-frame and converged-level comparisons on the reference's own frames; the
statement about WHEN the equaliser locks lives here.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch
from scipy import stats

import oracle
from conftest import load_golden
from vae_equalizer_amd import channel as ch
from vae_equalizer_amd import shared_funcs as sfun
from vae_equalizer_amd.dp_runs import host_threads
from vae_equalizer_amd.engine import DPEngine, dp_epilogue_compact

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PHI = np.array([0.0314, 0.0314], dtype=np.complex64)
TAU_PMD = 0.1e-12 * np.sqrt(1000)
M, B, SPS, LR, N_LRHALF = 25, 100, 2, 2.5e-3, 170


def run_ensemble(K, nu, SNR, F, N, theta_diff, seed, flex_step=None):
    """K runs x F frames x N symbols: (SER[K,4,F], Var_est[K,2,F]) of the HIP path and of the oracle on the same frames.
    flex_step: VAEflex (func_VAEflex_DP_MQAM_shaping.py:37-84: windows of B symbols advancing by flex_step, the centre flex_step outputs kept)."""
    t = sfun.qam_tables("64-QAM", nu)
    h_ch = sfun.upsampled_channel("h0", SPS)
    amps, P, nu_sc = t["amps"], t["P"], float(t["nu_sc"])
    var = float(t["pow_mean"] / 10 ** (SNR / 10) / 2)
    n = len(amps)
    if flex_step:
        N_out = (N - B) // flex_step * flex_step
        steps, stride, k0, klen, bl = N_out // flex_step, flex_step, (B - flex_step) // 2, flex_step, None
    else:
        N_out, steps, stride, k0, klen, bl = N, N // B, B, 0, B, B
    amp32 = amps.astype(np.float32)
    # product side
    eng = DPEngine(K, M, amps, P, [var, var], nu_sc, DEV, SPS)
    amp_t = torch.tensor(amp32, device=DEV)
    nu_t, var_t = torch.full((K,), nu_sc, device=DEV), torch.full((K, 2), var, device=DEV)
    # oracle side: the Dirac start of shared_funcs.py:495,583-586, zeroed Adam state
    W, h = np.zeros((K, 2, 4, M), np.float32), np.zeros((K, 2, 2, 2, M), np.float32)
    W[:, 0, 0, M // 2] = W[:, 1, 1, M // 2] = 1
    h[:, 0, 0, 0, M // 2] = h[:, 1, 1, 0, M // 2] = 1
    mW, vW, mh, vh, step = np.zeros_like(W), np.zeros_like(W), np.zeros_like(h), np.zeros_like(h), np.zeros(K, np.int32)
    Pk, vark, nuk = np.tile(P.astype(np.float32), (K, 1)), np.full((K, 2), var, np.float32), np.full(K, nu_sc, np.float32)
    q, y = np.zeros((K, 2, 2 * n, N_out), np.float32), np.zeros((K, 2, 2, N_out), np.float32)
    loss, ve = np.zeros((K, steps), np.float32), np.zeros((K, 2, steps), np.float32)
    cores = host_threads()
    pool = ThreadPoolExecutor(cores)
    SER = np.zeros((2, K, 4, F), np.float32)
    VE = np.zeros((2, K, 2, F), np.float32)
    for f in range(F):
        lr_W = LR * 0.5 if f >= N_LRHALF else LR                                # group 0 only (func_VAELE_DP_MQAM_shaping.py:45-46)
        rx, data = ch.generate_batch_hip(K, N, amps, P, SNR, h_ch, 90e9, SPS, -26e-24, TAU_PMD, PHI, np.pi / 10 + f * theta_diff, DEV, seed, f)
        if flex_step:
            data = data[:, :, :, B // 2:N_out + B // 2].contiguous()             # func_VAEflex_DP_MQAM_shaping.py:51
        out = eng.train(rx, B, steps, lr_W, LR, stride=stride, keep_off=k0, keep_len=klen, want_q=False, want_compact=True)
        res = dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp_t, nu_t, var_t, bl)
        SER[0, :, :, f] = res["SER"].cpu().numpy()
        VE[0, :, :, f] = out["var_est"][:, 0].mean(dim=2).cpu().numpy()        # :69
        rx_h, data_h = np.ascontiguousarray(rx.cpu().numpy()), data.cpu().numpy()
        lrw, lrh = np.full(K, lr_W, np.float32), np.full(K, LR, np.float32)
        oracle.dp_train_batch_f32(K, cores, steps, B, SPS, M, n, stride, k0, klen, rx_h, W, h, mW, vW, mh, vh, step, amp32, Pk, vark, nuk, lrw, lrh, q, y, loss, ve)
        VE[1, :, :, f] = ve.mean(axis=2)
        sers = list(pool.map(lambda i: oracle.dp_frame_epilogue(q[i], y[i], data_h[i], amp32, nu_sc, vark[i], bl)["SER"], range(K)))
        SER[1, :, :, f] = np.stack(sers)
    pool.shutdown()
    return SER, VE, var


def _log(*a):
    """Statistics of a run of these tests, kept when VAEQ_ENSEMBLE_LOG names a file (profiles/r03_*/ensemble_parity.txt is such a run)."""
    import os
    print(*a)
    if os.environ.get("VAEQ_ENSEMBLE_LOG"):
        with open(os.environ["VAEQ_ENSEMBLE_LOG"], "a") as fh:
            print(*a, file=fh)


def escape_frames(SER):
    """First frame from which on all four SER estimates stay below 0.1 for at least three frames (F if never)."""
    K, _, F = SER.shape
    ok = (SER < 0.1).all(axis=1)
    out = np.full(K, F)
    for k in range(K):
        for f in range(F - 2):
            if ok[k, f] and ok[k, f + 1] and ok[k, f + 2]:
                out[k] = f
                break
    return out


def converged_means(X, esc, F, margin=4, tail=None):
    """Per run: mean over the frames after its own escape (+ margin) -> [K, rows]; NaN rows for runs that never locked."""
    out = np.full((X.shape[0], X.shape[1]), np.nan)
    for k in range(X.shape[0]):
        lo = esc[k] + margin
        if lo < F - 4:
            out[k] = X[k, :, lo:].mean(axis=1)
    return out


def mc_agree(a, b, nsig=3.0, floor=0.0):
    """Ensemble means of two [K, rows] tables within nsig standard errors of their difference (NaN runs dropped), per row."""
    a, b = a[~np.isnan(a).any(1)], b[~np.isnan(b).any(1)]
    se = np.sqrt(a.var(axis=0, ddof=1) / len(a) + b.var(axis=0, ddof=1) / len(b))
    d = np.abs(a.mean(axis=0) - b.mean(axis=0))
    return d, nsig * se + floor


@pytest.mark.parametrize("case", ["pcs_G10", "vaele_G7", "flex_G9"])
def test_locking_runs_ensemble_vs_oracle(case):
    """Light shaping / uniform 64-QAM at 23 dB: every run locks.  G10 = config 5's PCS shape (200 frames x 3000 symbols, nu = 0.0270955), G7 = the
    VAE-LE trajectory capture (140 frames x 1000 symbols, nu = 0), G9 = config 4, VAEflex (70 frames x 2000 symbols = 13 300 window steps per run)."""
    flex_step = None
    if case == "flex_G9":
        g = load_golden("G9_flex_run")
        nu, F, N, td, ref_SER, ref_VE, K = 0.0, int(g["num_frames"]), int(g["N_frame_max"]), float(g["theta_diff"]), g["SER"], g["Var_est"], 32
        min_locked, flex_step = 0.9, 10
    elif case == "pcs_G10":
        g = load_golden("G10_pcs_run")
        nu, F, N, td, ref_SER, ref_VE, K = float(g["nu"]), int(g["num_frames"]), int(g["N_frame_max"]), float(g["theta_diff"]), g["SER"], g["Var_est"], 48
        min_locked = 0.9                                                        # 200 frames: (nearly) every run has locked by the end
    else:
        g = load_golden("G7_runs")
        nu, F, N, td, ref_SER, ref_VE, K = 0.0, int(g["vaele_num_frames"]), int(g["vaele_N_frame_max"]), float(g["vaele_theta_diff"]), g["vaele_SER"], g["vaele_Var_est"], 64
        min_locked = 0.6          # the capture ends at frame 140, inside the escape distribution (oracle: 108 .. beyond 140, the reference's run: 118): censored at F
    SER, VE, var = run_ensemble(K, nu, 23.0, F, N, td, seed=20260 + len(case), flex_step=flex_step)
    hip, orc = SER[0], SER[1]
    # (a) same frames, before chaos: frame 0 run by run (N / 100 steps from the Dirac start; a VAEflex frame is already 190 window steps, past the
    #     horizon of run-by-run agreement: there the frame's mean noise estimate within 5 %)
    tol_ve, tol_ser = (5e-2, 0.05) if flex_step else (2e-3, 0.02)
    assert np.max(np.abs(VE[0][:, :, 0] - VE[1][:, :, 0]) / VE[1][:, :, 0]) < tol_ve
    assert np.max(np.abs(hip[:, :, 0] - orc[:, :, 0])) < tol_ser
    # (b) escape frames: same distribution
    eh, eo = escape_frames(hip), escape_frames(orc)
    ks = stats.ks_2samp(eh, eo)
    info = dict(hip=(int(eh.min()), float(np.median(eh)), int(eh.max())), oracle=(int(eo.min()), float(np.median(eo)), int(eo.max())), ks_p=float(ks.pvalue))
    _log(case, f"K={K} runs x {F} frames x {N} symbols; escape frames (min, median, max):", info, "locked fraction hip/oracle:", float((eh < F).mean()), float((eo < F).mean()))
    assert ks.pvalue > 0.01, info
    assert abs(np.median(eh) - np.median(eo)) <= 0.15 * np.median(eo), info
    assert min((eh < F).mean(), (eo < F).mean()) >= min_locked and abs((eh < F).mean() - (eo < F).mean()) <= 0.1, info   # the same fraction locks on both sides
    # (c) converged level: SER (4 estimators) and noise estimate, ensemble means within 3 sigma of the Monte-Carlo error
    d, lim = mc_agree(converged_means(hip, eh, F), converged_means(orc, eo, F))
    _log(case, "converged SER |hip - oracle| per estimator:", d, "3 sigma:", lim, "level:", np.nanmean(converged_means(orc, eo, F), axis=0))
    assert np.all(d <= lim), (d, lim)
    d, lim = mc_agree(converged_means(VE[0], eh, F), converged_means(VE[1], eo, F))
    _log(case, "converged Var_est |hip - oracle|:", d, "3 sigma:", lim, "level:", np.nanmean(converged_means(VE[1], eo, F), axis=0), "true var:", var)
    assert np.all(d <= lim), (d, lim)
    # (d) the reference's captured run (its own seed, host generator) inside the pooled ensemble's range
    er = escape_frames(ref_SER[None])[0]
    pooled = np.concatenate([eh, eo])
    _log(case, "the reference's captured run: escape frame", int(er), "pooled ensemble range", int(pooled.min()), int(pooled.max()))
    assert pooled.min() <= er <= pooled.max(), (int(er), info)
    cm = np.concatenate([converged_means(hip, eh, F), converged_means(orc, eo, F)])
    cm = cm[~np.isnan(cm).any(1)]
    rm = converged_means(ref_SER[None], np.array([er]), F)[0]
    assert np.all(rm >= cm.min(axis=0) - 1e-3) and np.all(rm <= cm.max(axis=0) + 1e-3), (rm, cm.min(axis=0), cm.max(axis=0))
    cv = np.concatenate([converged_means(VE[0], eh, F), converged_means(VE[1], eo, F)])
    cv = cv[~np.isnan(cv).any(1)]
    rv = converged_means(ref_VE[None], np.array([er]), F)[0]
    assert np.all(rv >= 0.97 * cv.min(axis=0)) and np.all(rv <= 1.03 * cv.max(axis=0)), (rv, cv.min(axis=0), cv.max(axis=0))


@pytest.mark.parametrize("name", ["G13_cfg5_nu0872_snr20", "G13_cfg5_nu1222_snr28"])
def test_heavy_shaping_plateau_ensemble_vs_oracle(name):
    """Config 5's heavy-shaping points (Eval_run_DP.py:24,34; H = 4.6 / 4.125 bit): the reference's blind equaliser does not lock within the run.
    An unlocked equaliser wanders on its plateau, chaotically run by run -- as an ensemble, the plateau's SER and noise estimate are the same on
    the HIP path and in the oracle (window by window, 3 sigma of the Monte-Carlo error), and the reference's captured run lies inside it."""
    g = load_golden(name)
    F, N, K = int(g["num_frames"]), int(g["N_frame_max"]), 24
    SER, VE, var = run_ensemble(K, float(g["nu"]), float(g["SNR"]), F, N, float(g["theta_diff"]), seed=77 + int(float(g["SNR"])))
    hip, orc = SER[0], SER[1]
    assert np.max(np.abs(VE[0][:, :, 0] - VE[1][:, :, 0]) / VE[1][:, :, 0]) < 2e-2               # frame 0 (logits ~1e3 .. 1e4: the touchiest points)
    assert np.max(np.abs(hip[:, :, 0] - orc[:, :, 0])) < 0.05
    lock_h, lock_o = (hip[:, :, -20:] < 0.1).all(axis=(1, 2)), (orc[:, :, -20:] < 0.1).all(axis=(1, 2))
    assert lock_h.sum() <= 1 and lock_o.sum() <= 1, (lock_h.sum(), lock_o.sum())                 # (nearly) no run locks, on either side
    keep_h, keep_o = ~lock_h, ~lock_o
    for a, b in ((20, 60), (60, 120), (120, 200)):
        wh, wo = hip[keep_h][:, :, a:b].mean(axis=2), orc[keep_o][:, :, a:b].mean(axis=2)         # [K, 4] plateau SER per run
        d, lim = mc_agree(wh, wo)
        _log(name, f"frames {a}..{b}: plateau SER |hip - oracle|", d, "3 sigma", lim, "level", wo.mean(axis=0), "reference", g["SER"][:, a:b].mean(axis=1))
        assert np.all(d <= lim), ("SER", a, b, d, lim)
        vh_, vo_ = VE[0][keep_h][:, :, a:b].mean(axis=2), VE[1][keep_o][:, :, a:b].mean(axis=2)
        d, lim = mc_agree(vh_, vo_)
        _log(name, f"frames {a}..{b}: plateau Var_est |hip - oracle|", d, "3 sigma", lim, "level", vo_.mean(axis=0), "reference", g["Var_est"][:, a:b].mean(axis=1))
        assert np.all(d <= lim), ("Var_est", a, b, d, lim)
        # the reference's run inside the pooled range (SER rows within 0.01, the noise estimate within 3 %)
        ps, pv = np.concatenate([wh, wo]), np.concatenate([vh_, vo_])
        rs, rv = g["SER"][:, a:b].mean(axis=1), g["Var_est"][:, a:b].mean(axis=1)
        assert np.all(rs >= ps.min(axis=0) - 0.01) and np.all(rs <= ps.max(axis=0) + 0.01), (a, b, rs, ps.min(axis=0), ps.max(axis=0))
        assert np.all(rv >= 0.97 * pv.min(axis=0)) and np.all(rv <= 1.03 * pv.max(axis=0)), (a, b, rv, pv.min(axis=0), pv.max(axis=0))
    assert VE[0][keep_h][:, :, 100:].mean() > 4 * var                                            # the plateau's noise estimate, far above the true variance


def test_awgn_config2_ensemble_vs_oracle():
    """Config 2 (AWGN 64-QAM + PCS nu = 0.0270955, h1, 24 dB, 25 taps, lr 5e-3, minibatches of 350, 1200 symbols per epoch,
    Eval_run_shaping_vaele.py:19-36): WHEN the blind equaliser locks is chaotic from the very first step (the Dirac start's loss is invariant to the
    scale of its single tap: the first Adam step moves it by +-lr on rounding noise; the oracle locks at validation 47 ... 143 on ONE set of frames
    depending on precision).  As an ensemble -- K runs through vaeq_awgn_train / vaeq_awgn_validate and, epoch by epoch on the same frames, through
    the f32 oracle (func_VAELE_MQAM_shaping.py:291-322) -- the lock-epoch distributions and the converged SER agree, and the reference's run lies inside."""
    from vae_equalizer_amd.engine import AWGNEngine
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    g = load_golden("G7_awgn_cfg2")
    K, EPOCHS, EPE, NV, NT, Bm, Mt, lr = 96, 400, 4, 6000, 1200, 350, 25, 5e-3
    t = awgn_tables("64-QAM", 0.0270955, 24, "h1", SPS)
    amp32, P32 = t["amps"].astype(np.float32), t["P"].astype(np.float32)
    snr = np.full(K, 24, np.float32)
    eng = AWGNEngine(K, Mt, t["amps"], np.tile(t["P"], (K, 1)), t["amp_mean"], t["var"], DEV, SPS)
    states = [oracle.AWGNState(Mt, np.float32) for _ in range(K)]
    pool = ThreadPoolExecutor(host_threads())
    nval = EPOCHS // EPE
    SER = np.zeros((2, K, nval), np.float32)
    draw = 0
    for ep in range(EPOCHS):
        rx, _ = ch.generate_awgn_batch_hip(K, NT, t["amps"], t["P"], snr, t["h_channel"], SPS, DEV, 4242, draw); draw += 1
        eng.train(rx, Bm, NT // Bm, lr)
        rx_h = rx.cpu().numpy()
        list(pool.map(lambda i: oracle.awgn_train(states[i], rx_h[i], NT // Bm, Bm, amp32, P32, float(t["amp_mean"]), float(t["var"]), lr, SPS), range(K)))
        if ep % EPE == 0:                                                        # :308-318
            rxv, dv = ch.generate_awgn_batch_hip(K, NV, t["amps"], t["P"], snr, t["h_channel"], SPS, DEV, 4242, draw); draw += 1
            ser, _, _ = eng.validate(rxv, dv, 21)
            SER[0, :, ep // EPE] = ser.cpu().numpy()
            xv, dvh = rxv.cpu().numpy(), dv.cpu().numpy()

            def val(i):
                q, _ = oracle.awgn_forward(xv[i], states[i].W, amp32, float(t["amp_mean"]), float(t["var"]), SPS)
                return oracle.awgn_validate(q, dvh[i], amp32)[0]
            SER[1, :, ep // EPE] = list(pool.map(val, range(K)))
    pool.shutdown()
    lock = lambda S: np.array([int(np.argmax(s < 0.01)) if (s < 0.01).any() else nval for s in S]) * EPE      # epoch of the first validation below 1 %
    lh, lo = lock(SER[0]), lock(SER[1])
    ks = stats.ks_2samp(lh, lo)
    ref_lock = int(np.argmax(g["SER"] < 0.01)) * 2                              # the capture validates every 2nd epoch
    info = dict(hip=(int(lh.min()), float(np.median(lh)), int(lh.max())), oracle=(int(lo.min()), float(np.median(lo)), int(lo.max())), ks_p=float(ks.pvalue), reference=ref_lock)
    _log("awgn_cfg2", f"K={K} runs x {EPOCHS} epochs; lock epoch (min, median, max):", info)
    # The distribution is BIMODAL: the noise-driven first step of the scale tap sends a run either towards a fast lock (epoch ~100) or a slow one
    # (~270-330) -- a rounding-level coin flip per run (the oracle in fp64 on the reference's frames: 57 / 132 / 143 depending on that sign).  So:
    # the same two modes at the same places, mixture weights within the binomial error of K runs, and the two-sample KS test on the whole.
    assert ks.pvalue > 0.01, info
    split = 180
    fh, fo = (lh < split).mean(), (lo < split).mean()
    _log("awgn_cfg2", "fast-mode fraction hip / oracle:", float(fh), float(fo), "fast medians:", float(np.median(lh[lh < split])), float(np.median(lo[lo < split])),
         "slow medians:", float(np.median(lh[lh >= split])), float(np.median(lo[lo >= split])))
    assert abs(fh - fo) <= 3 * np.sqrt(0.25 * 2 / K), (fh, fo)
    assert 0.15 < fh < 0.85 and 0.15 < fo < 0.85, (fh, fo)                      # both modes populated on both sides
    assert abs(np.median(lh[lh < split]) - np.median(lo[lo < split])) <= 0.15 * np.median(lo[lo < split]), info
    assert abs(np.median(lh[lh >= split]) - np.median(lo[lo >= split])) <= 0.2 * np.median(lo[lo >= split]), info
    assert (lh < EPOCHS).mean() >= 0.85 and (lo < EPOCHS).mean() >= 0.85, info
    pooled = np.concatenate([lh, lo])
    assert pooled.min() - 8 <= ref_lock <= pooled.max(), info                   # (validation grids differ: every 4th epoch here, every 2nd in the capture)
    # converged SER: the last 15 validations of the runs locked by then
    th, to = SER[0][lh < EPOCHS - 20 * EPE][:, -15:].mean(axis=1), SER[1][lo < EPOCHS - 20 * EPE][:, -15:].mean(axis=1)
    se = np.sqrt(th.var(ddof=1) / len(th) + to.var(ddof=1) / len(to))
    _log("awgn_cfg2", "converged SER hip / oracle / reference:", float(th.mean()), float(to.mean()), float(g["SER"][-50:].mean()), "3 sigma:", float(3 * se))
    assert abs(th.mean() - to.mean()) <= 3 * se + 2e-5, (th.mean(), to.mean(), se)
    assert abs(th.mean() - g["SER"][-50:].mean()) < 3e-4
