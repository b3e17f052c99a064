"""HIP channel generator (vaeq_gen_dp_*, row f1) against the numpy restatement of the reference's generator chain, fed with the
same Philox symbol stream (reproduced on the host), plus noise statistics."""
import os

import numpy as np
import pytest
import torch

from vae_equalizer_amd import channel as ch
from vae_equalizer_amd import shared_funcs as sfun

pytestmark = pytest.mark.gpu
DP = dict(symb_rate=90e9, tau_cd=-26e-24, tau_pmd=0.1e-12 * np.sqrt(1000), phiIQ=np.array([0.0314, 0.0314], dtype=np.complex64))
M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """numpy Philox4x32-10 (Salmon et al.), vectorised over uint64 arrays holding 32-bit words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & M32 for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & M32, np.uint64(k1) & M32
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M32, (k1 + np.uint64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def host_symbols(seed, frame, run, pol, n_idx, cdf):
    """Host replica of draw_symbol_pair (vaeq_gen.hip): Philox counter n >> 1, words (x, y) for even n and (z, w) for odd n."""
    key = ch._mix_seed(seed, 0)
    x, y, z, w = philox4x32_10(n_idx >> 1, run, frame, pol, key & 0xFFFFFFFF, key >> 32)
    odd = (n_idx & 1).astype(bool)
    u = lambda v: ((v >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    lev = lambda uu: (uu[:, None] >= cdf[None, :-1].astype(np.float32)).sum(1)
    return lev(u(np.where(odd, z, x))), lev(u(np.where(odd, w, y)))


@pytest.mark.parametrize("mod,nu,channel", [("64-QAM", 0.0270955, "h0"), ("16-QAM", 0.0, "h1")])
def test_hip_generator_matches_numpy_chain(mod, nu, channel):
    sps, N, R, seed, frame = 2, 600, 3, 77, 5
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init(channel, mod, "cpu", nu, sps, 25, 23)
    theta = np.array([0.3, 0.9, -0.4])
    SNR = np.array([23.0, 18.0, 30.0], np.float32)
    rx, data, sigma = ch.generate_batch_hip(R, N, amps, P, SNR, h_ch, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], theta,
                                            "cuda:0", seed, frame, return_sigma=True, fft="exact")
    rx2, data2 = ch.generate_batch_hip(R, N, amps, P, SNR, h_ch, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], theta,
                                       "cuda:0", seed, frame, fft="exact")
    assert torch.equal(rx, rx2) and torch.equal(data, data2)                    # deterministic in (seed, frame, run)
    geo = ch.dp_frame_geometry(N, h_ch, sps)
    cdf = np.cumsum(P)
    amps32 = np.asarray(amps, np.float32)
    for r in range(R):
        lev = np.stack([np.stack(host_symbols(seed, frame, r, p, np.arange(geo["N_conv"]), cdf)) for p in range(2)])   # [pol][I/Q][n]
        sym = amps32[lev]
        lo = geo["ref_offset"]
        assert np.array_equal(data[r].cpu().numpy(), sym[:, :, lo:lo + N].astype(np.float16))       # TX reference (:89)
        tx_up = np.zeros((2, sps * (geo["N_conv"] - 1) + 1), np.complex64)
        tx_up[:, ::sps] = sym[:, 0] + 1j * sym[:, 1]
        clean = ch.simulate_dispersion(ch.simulate_channel(tx_up, ch.rrcfir(8, sps, 0.1), h_ch), DP["symb_rate"], sps, DP["tau_cd"],
                                       DP["tau_pmd"], DP["phiIQ"], theta[r])                         # (:80-81)
        sig_n = np.sqrt(np.mean(np.abs(clean) ** 2) * sps / 2 / 10 ** (SNR[r] / 10))                 # (:83)
        assert abs(float(sigma[r]) - sig_n) / sig_n < 1e-4
        got = rx[r].cpu().numpy()
        noise = np.stack([got[:, 0] - clean[:, :sps * N].real, got[:, 1] - clean[:, :sps * N].imag], 1)   # [pol][I/Q][s]
        assert abs(noise.std() / sig_n - 1) < 0.05 and abs(noise.mean()) < 4 * sig_n / np.sqrt(noise.size)
        assert abs(np.corrcoef(noise[:, 0].ravel(), noise[:, 1].ravel())[0, 1]) < 0.08
        assert np.abs(noise).max() < 6 * sig_n                                                        # the clean part matches to << sigma
    # different frames / seeds give different data
    rx3, _ = ch.generate_batch_hip(R, N, amps, P, SNR, h_ch, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], theta, "cuda:0", seed, frame + 1,
                                   fft="exact")
    assert not torch.equal(rx, rx3)


def test_hip_generator_clean_signal_accuracy():
    """With SNR = 200 dB the noise vanishes: rx == numpy chain to c64-FFT accuracy."""
    sps, N, seed, frame = 2, 512, 3, 0
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h1", "64-QAM", "cpu", 0.0872449, sps, 25, 23)
    rx, data = ch.generate_batch_hip(1, N, amps, P, 200.0, h_ch, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], 0.7, "cuda:0", seed, frame,
                                      fft="exact")
    geo = ch.dp_frame_geometry(N, h_ch, sps)
    lev = np.stack([np.stack(host_symbols(seed, frame, 0, p, np.arange(geo["N_conv"]), np.cumsum(P))) for p in range(2)])
    sym = np.asarray(amps, np.float32)[lev]
    tx_up = np.zeros((2, sps * (geo["N_conv"] - 1) + 1), np.complex64)
    tx_up[:, ::sps] = sym[:, 0] + 1j * sym[:, 1]
    clean = ch.simulate_dispersion(ch.simulate_channel(tx_up, ch.rrcfir(8, sps, 0.1), h_ch), DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"],
                                   DP["phiIQ"], 0.7)
    got = rx[0].cpu().numpy()
    ref = np.stack([clean[:, :sps * N].real, clean[:, :sps * N].imag], 1)
    assert np.max(np.abs(got - ref)) < 2e-5 * np.max(np.abs(ref)) + 1e-5


def test_hip_generator_padded_fft_differs_only_at_the_frame_edges():
    """fft="padded" (fast FFT length, linear filtering) vs fft="exact" (the reference's circular filtering over Ls): same symbols,
    same noise; the clean signals differ only within the dispersion's impulse-response length of the frame edges."""
    sps, N, seed, frame = 2, 4000, 11, 2
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", "64-QAM", "cpu", 0.0, sps, 25, 23)
    args = (2, N, amps, P, 200.0, h_ch, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], np.array([0.3, 1.1]), "cuda:0", seed, frame)
    rxe, de, se = ch.generate_batch_hip(*args, fft="exact", return_sigma=True)
    rxp, dp_, sp = ch.generate_batch_hip(*args, fft="padded", return_sigma=True)
    assert torch.equal(de, dp_)
    # the noise level comes from the power BEFORE the (unitary) fibre: identical in both modes, and exact for the circular transform over Ls
    # ("exact").  In "padded" mode a little energy of the linear filtering lands in the zero pad, so the power of what is kept differs from it:
    # bound that against the post-dispersion power of the returned (noise-free: 200 dB) samples -- 2 x 8000 samples estimate it to ~1 %
    assert torch.equal(se, sp)
    for rx_, tag in ((rxe, "exact"), (rxp, "padded")):
        p_post = (rx_.double() ** 2).sum(dim=2).mean(dim=(1, 2)).cpu().numpy()                    # mean |sig|^2 per run over both polarisations
        p_pre = (se.double().cpu().numpy() ** 2) * 2 / sps * 10 ** (200.0 / 10)                     # sigma^2 = P sps / 2 / 10^(SNR/10)   (:83)
        assert np.all(np.abs(p_post / p_pre - 1) < 0.03), (tag, p_post / p_pre)
    e, p = rxe.cpu().numpy(), rxp.cpu().numpy()
    scale = np.abs(e).max()
    assert np.max(np.abs(e - p)[..., 64:-64]) < 1e-3 * scale                   # interior: equal up to the tails of the response
    assert np.max(np.abs(e - p)) < 0.5 * scale                                 # edges: wrapped-around vs. absent neighbours
    assert ch.fast_fft_len(20034 + 64) == 20480 and ch.fast_fft_len(1024) == 1024 and ch.fast_fft_len(1025) == 1280
    assert ch.padded_row_len(20034 + 64) == 20480 and ch.padded_row_len(2098) == 4096 and ch.padded_row_len(6098) == 8192 and ch.padded_row_len(40098) == 40960


@pytest.mark.parametrize("n1", [4, 5, 8, 10, 16, 20])
def test_fused_frame_matches_staged_chain(n1, monkeypatch):
    """vaeq_gen_dp_frame's three-pass form (pulse shaping + outer DFT stage | per-row 1024-point FFT, fibre matrix, inverse FFT | inverse outer
    stage + noise) against the five-pass hipFFT chain it replaces: same symbols and TX reference bit for bit, the same noise words, the clean
    signal to transform rounding -- for every row length N1 * 1024 the fused form covers, with per-run rotation angles, SNRs and shaping."""
    sps, seed, frame, R = 2, 5, 3, 6
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h1", "64-QAM", "cpu", 0.0270955, sps, 25, 23)
    N = (1024 * n1 - 64 - ch.dp_frame_geometry(100, h_ch, sps)["Ls"] + 200) // 2       # the longest frame whose padded row is N1 * 1024
    geo = ch.dp_frame_geometry(N, h_ch, sps)
    assert ch.padded_row_len(geo["Ls"] + 64) == 1024 * n1 and geo["Ls"] + 64 >= 1024 * n1 - 1
    Pr = np.stack([P if r % 2 == 0 else np.full_like(P, 1 / len(P)) for r in range(R)])
    theta = np.linspace(-1.2, 2.9, R)
    SNR = np.linspace(14.0, 30.0, R).astype(np.float32)
    args = (R, N, amps, Pr, SNR, h_ch, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], theta, "cuda:0", seed, frame)
    monkeypatch.delenv("VAEQ_GEN_STAGED", raising=False)
    rxf, df, sf = ch.generate_batch_hip(*args, return_sigma=True)
    rxf2, _ = ch.generate_batch_hip(*args)
    assert torch.equal(rxf, rxf2)                                              # deterministic
    monkeypatch.setenv("VAEQ_GEN_STAGED", "1")
    rxs, ds, ss = ch.generate_batch_hip(*args, return_sigma=True)
    assert torch.equal(df, ds)
    assert torch.allclose(sf, ss, rtol=2e-6, atol=0)
    scale = float(rxs.abs().max())
    assert float((rxf - rxs).abs().max()) < 1e-5 * scale
    # noiseless: the clean signals alone agree to transform rounding
    args200 = args[:4] + (200.0,) + args[5:]
    clean_s, _ = ch.generate_batch_hip(*args200)
    monkeypatch.delenv("VAEQ_GEN_STAGED")
    clean_f, _ = ch.generate_batch_hip(*args200)
    assert float((clean_f - clean_s).abs().max()) < 1e-5 * float(clean_s.abs().max())
    assert float((clean_f - clean_s).abs().mean()) < 1e-6 * float(clean_s.abs().max())


def test_fused_frame_long_pulse_and_last_stripe_ownership():
    """A combined pulse longer than 64 taps: symbols past Lrow / 2 exist and are owned by the last stripe's halo; chunked runs keep their streams."""
    sps, seed, frame, R = 2, 9, 1, 3
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h1", "16-QAM", "cpu", 0.0, sps, 25, 23)
    h_long = np.concatenate([np.asarray(h_ch), 0.05 * np.exp(1j * np.arange(50))]).astype(np.complex64)
    N = (1024 * 4 - 64 - ch.dp_frame_geometry(100, h_long, sps)["Ls"] + 200) // 2
    geo = ch.dp_frame_geometry(N, h_long, sps)
    assert geo["Lg"] > 64 and ch.padded_row_len(geo["Ls"] + 64) == 4096
    args = (R, N, amps, P, 21.0, h_long, DP["symb_rate"], sps, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"], np.array([0.1, 0.5, 2.0]), "cuda:0", seed, frame)
    os.environ.pop("VAEQ_GEN_STAGED", None)
    rxf, df = ch.generate_batch_hip(*args)
    os.environ["VAEQ_GEN_STAGED"] = "1"
    try:
        rxs, ds = ch.generate_batch_hip(*args)
    finally:
        os.environ.pop("VAEQ_GEN_STAGED", None)
    assert torch.equal(df, ds)
    assert float((rxf - rxs).abs().max()) < 1e-5 * float(rxs.abs().max())


@pytest.mark.parametrize("N,fixed", [(1200, False), (3900, False), (700, True), (1900, False)])
def test_awgn_generator_one_pass_equals_two_pass(N, fixed, monkeypatch):
    """Short AWGN frames (up to four 2048-sample tiles: the training frames of both AWGN scripts) are generated in one pass, one workgroup per run;
    the result is bit for bit the two-pass form's (power pass + noise pass, what longer frames use): same symbols, same power sums in the same
    order, same noise words."""
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    t = awgn_tables("64-QAM", 0.0270955, 24, "h1", 2)
    R = 37
    P = np.stack([t["P"] if r % 3 else np.full_like(t["P"], 1 / len(t["P"])) for r in range(R)])
    snr = np.linspace(12, 30, R).astype(np.float32)
    kw = dict(sigma_fixed=np.linspace(0.01, 0.2, R).astype(np.float32)) if fixed else {}
    monkeypatch.delenv("VAEQ_AWGN_TWOPASS", raising=False)
    a = ch.generate_awgn_batch_hip(R, N, t["amps"], P, snr, t["h_channel"], 2, "cuda:0", 5, 2, return_sigma=True, **kw)
    monkeypatch.setenv("VAEQ_AWGN_TWOPASS", "1")
    b = ch.generate_awgn_batch_hip(R, N, t["amps"], P, snr, t["h_channel"], 2, "cuda:0", 5, 2, return_sigma=True, **kw)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_generated_frames_train():
    """End to end: frames from the HIP generator make the equalizer converge (loss falls, Var_est falls)."""
    from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
    runs = [DPRun(23, 0.0, 0.0, 0.3, 2.5e-3, 90e9, seed=9) for _ in range(4)]
    r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 2000, 4, 10, "h0", -26e-24, DP["tau_pmd"], DP["phiIQ"], 170, generator="hip")
    assert torch.isfinite(r["SER"]).all() and (r["Var_est"][:, :, -1] < r["Var_est"][:, :, 0]).all()


# ------------------------------------------------------------------ AWGN / ISI channel generator (vaeq_gen_awgn)
def test_awgn_generator_statistics_and_structure():
    """vaeq_gen_awgn against the model of AWGN_channel/func_VAELE_MQAM_shaping.py:39-61: PCS pmf of the symbols, rx = (zero-stuffed
    symbols * rrc * h_channel) + white noise with sigma_n from the measured mean power, reference aligned at T + M_channel - 1."""
    import numpy as np
    import torch
    from vae_equalizer_amd import channel as ch
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    R, N, sps = 6, 6000, 2
    snr = np.array([10, 14, 18, 22, 26, 30], np.float32)
    t = awgn_tables("64-QAM", 0.0270955, 20, "h1", sps)
    rx, data, sigma = ch.generate_awgn_batch_hip(R, N, t["amps"], t["P"], snr, t["h_channel"], sps, "cuda:0", 1234, 3, return_sigma=True)
    rx2, data2 = ch.generate_awgn_batch_hip(R, N, t["amps"], t["P"], snr, t["h_channel"], sps, "cuda:0", 1234, 3)
    assert torch.equal(rx, rx2) and torch.equal(data, data2)                  # counter-based: reproducible
    rx3, _ = ch.generate_awgn_batch_hip(R, N, t["amps"], t["P"], snr, t["h_channel"], sps, "cuda:0", 1234, 4)
    assert not torch.equal(rx, rx3)
    assert rx.shape == (R, 2, sps * N) and data.shape == (R, 2, N) and data.dtype == torch.float16
    rx, data, sigma = rx.cpu().numpy().astype(np.float64), data.cpu().numpy().astype(np.float64), sigma.cpu().numpy()
    amps = np.asarray(t["amps"])
    lev = np.argmin(np.abs(data[..., None] - amps), axis=-1)
    assert np.max(np.abs(data - amps[lev])) < 1e-3                            # fp16 levels
    pmf = np.bincount(lev.reshape(-1), minlength=len(amps)) / lev.size
    assert np.max(np.abs(pmf - t["P"])) < 4 * np.sqrt(0.25 / lev.size)
    assert not np.array_equal(lev[0], lev[1]) and not np.array_equal(lev[0, 0], lev[0, 1])
    geo = ch.awgn_frame_geometry(N, t["h_channel"], sps)
    for r in range(R):
        up = np.zeros(sps * (N - 1) + 1, complex)
        up[::sps] = data[r, 0] + 1j * data[r, 1]
        clean = np.convolve(up, geo["g"].astype(complex), mode="valid")      # samples whose pulse support lies inside the reference
        o = sps * geo["ref_offset"]
        got = (rx[r, 0] + 1j * rx[r, 1])[o:o + len(clean)]
        noise = got - clean[:len(got)]
        p_sig = np.mean(np.abs(clean) ** 2)
        want_sigma = np.sqrt(sps * p_sig / 2 / 10 ** (snr[r] / 10))
        assert abs(sigma[r] / want_sigma - 1) < 0.03                          # mean power over the run's own (slightly longer) sequence
        assert abs(np.std(noise.real) / sigma[r] - 1) < 0.03 and abs(np.std(noise.imag) / sigma[r] - 1) < 0.03
        assert abs(np.mean(noise)) < 4 * sigma[r] / np.sqrt(len(noise))
        ac = np.abs(np.vdot(noise[1:], noise[:-1])) / np.vdot(noise, noise).real
        assert ac < 5 / np.sqrt(len(noise))                                   # white


def test_awgn_generator_generic_oversampling():
    """sps = 3 takes the generic stage-1 / reference kernels (one thread per sample): same structure checks as for sps = 2."""
    import numpy as np
    from vae_equalizer_amd import channel as ch
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    R, N, sps = 3, 3000, 3
    snr = np.array([12, 20, 28], np.float32)
    t = awgn_tables("16-QAM", 0.0, 20, "h2", sps)
    rx, data, sigma = ch.generate_awgn_batch_hip(R, N, t["amps"], t["P"], snr, t["h_channel"], sps, "cuda:0", 77, 0, return_sigma=True)
    assert rx.shape == (R, 2, sps * N) and data.shape == (R, 2, N)
    rx, data, sigma = rx.cpu().numpy().astype(np.float64), data.cpu().numpy().astype(np.float64), sigma.cpu().numpy()
    geo = ch.awgn_frame_geometry(N, t["h_channel"], sps)
    for r in range(R):
        up = np.zeros(sps * (N - 1) + 1, complex)
        up[::sps] = data[r, 0] + 1j * data[r, 1]
        clean = np.convolve(up, geo["g"].astype(complex), mode="valid")
        o = sps * geo["ref_offset"]
        got = (rx[r, 0] + 1j * rx[r, 1])[o:o + len(clean)]
        noise = got - clean[:len(got)]
        want_sigma = np.sqrt(sps * np.mean(np.abs(clean) ** 2) / 2 / 10 ** (snr[r] / 10))
        assert abs(sigma[r] / want_sigma - 1) < 0.04
        assert abs(np.std(noise.real) / sigma[r] - 1) < 0.04 and abs(np.std(noise.imag) / sigma[r] - 1) < 0.04
