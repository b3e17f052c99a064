"""Channel generators: the numpy restatement reproduces the reference's frames under the same seeds (G6);
the batched torch generator has the same statistics."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from vae_equalizer_amd import channel as ch
from vae_equalizer_amd import shared_funcs as sfun

DP = dict(symb_rate=90e9, tau_cd=-26e-24, tau_pmd=0.1e-12 * np.sqrt(1000), phiIQ=np.array([0.0314, 0.0314], dtype=np.complex64))


def test_pulse_filters():
    g = load_golden("G6_generator")
    assert np.max(np.abs(ch.rrcfir(8, 2, 0.1) - g["rrc_8_2_01"])) < 1e-7
    assert np.max(np.abs(ch.rcfir(8, 2, 0.1) - g["rc_8_2_01"])) < 1e-7


def test_dp_generator_matches_reference_frames():
    g = load_golden("G6_generator")
    for tag in ("a", "b"):
        mod, nu, SNR, channel = g[f"{tag}_args"]
        h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init(str(channel), str(mod), "cpu", float(nu), 2, 25, float(SNR))
        st = ch.SeededStreams(61)
        rx, data, sig = ch.generate_data_shaping(300, amps, float(SNR), h_ch, P, pol, DP["symb_rate"], 2, DP["tau_cd"], DP["tau_pmd"],
                                                 DP["phiIQ"], 0.3, "cpu", rng=st.next_rng(), noise=st.noise)
        rx2, data2, _ = ch.generate_data_shaping(300, amps, float(SNR), h_ch, P, pol, DP["symb_rate"], 2, DP["tau_cd"], DP["tau_pmd"],
                                                 DP["phiIQ"], 0.5, "cpu", rng=st.next_rng(), noise=st.noise)
        assert rx.shape == (2, 2, 600) and rx.dtype == torch.float32 and data.dtype == torch.float16
        assert np.array_equal(data.numpy(), g[f"{tag}_data"]) and np.array_equal(data2.numpy(), g[f"{tag}_data2"])
        assert np.max(np.abs(rx.numpy() - g[f"{tag}_rx"])) < 2e-6
        assert np.max(np.abs(rx2.numpy() - g[f"{tag}_rx2"])) < 2e-6
        assert abs(sig - float(g[f"{tag}_sigma_n"])) < 1e-7


def test_awgn_generator_matches_reference_frame():
    g = load_golden("G6_generator")
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    t = awgn_tables("16-QAM", 0.0, 24, "h1", 2)
    st = ch.SeededStreams(62)
    rx, data = ch.generate_data(200, t["M_channel"], t["amps"], 24, t["h_channel"], 2, "cpu", t["P"], rng=st.next_rng(), noise=st.noise)
    assert np.array_equal(data.numpy(), g["awgn_data"])
    assert np.max(np.abs(rx.numpy() - g["awgn_rx"])) < 2e-6


def test_batched_generator_statistics():
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", "64-QAM", "cpu", 0.0270955, 2, 25, 23)
    gen = torch.Generator().manual_seed(1)
    rx, data = ch.generate_batch_gpu(6, 2000, amps, P, 23.0, h_ch, DP["symb_rate"], 2, DP["tau_cd"], DP["tau_pmd"], DP["phiIQ"],
                                     np.linspace(0.1, 0.6, 6), "cpu", generator=gen)
    st = ch.SeededStreams(3)
    rx_ref, data_ref, _ = ch.generate_data_shaping(2000, amps, 23, h_ch, P, pol, DP["symb_rate"], 2, DP["tau_cd"], DP["tau_pmd"],
                                                   DP["phiIQ"], 0.3, "cpu", rng=st.next_rng(), noise=st.noise)
    assert rx.shape == (6, 2, 2, 4000) and data.shape == (6, 2, 2, 2000) and data.dtype == torch.float16
    assert abs(float(rx.std()) / float(rx_ref.std()) - 1) < 0.05
    assert abs(float(data.float().pow(2).mean()) / float(data_ref.float().pow(2).mean()) - 1) < 0.05
    # PCS: empirical level distribution follows P
    lev = torch.round((data.float() / float(amps[1] - amps[0])) + 3.5).long().clamp(0, 7)
    emp = torch.bincount(lev.flatten(), minlength=8).double() / lev.numel()
    assert np.max(np.abs(emp.numpy() - P)) < 0.01


def test_cdf_cache_is_keyed_by_content():
    """channel._cdf_dev: the runs' cumulative PCS tables are cached by content -- equal tables share one tensor, an in-place change misses."""
    import torch
    R, n = 5, 8
    P = np.random.default_rng(0).dirichlet(np.ones(n), R)
    a = ch._cdf_dev(P, R, n, "cpu")
    assert a.shape == (R, n) and a.dtype == torch.float32
    assert np.array_equal(a.numpy(), np.cumsum(P, axis=1).astype(np.float32))
    assert ch._cdf_dev(P.copy(), R, n, "cpu") is a
    P[2, 3] += 1e-9
    b = ch._cdf_dev(P, R, n, "cpu")
    assert b is not a
    row = ch._cdf_dev(P[0], R, n, "cpu")                      # one table for all runs
    assert row.shape == (R, n) and row.is_contiguous() and np.array_equal(row.numpy(), np.tile(np.cumsum(P[0]).astype(np.float32), (R, 1)))
    with pytest.raises(ValueError):
        ch._cdf_dev(P[:3], R, n, "cpu")
    g1, g2 = ch.dp_frame_geometry(600, [1.0, 0.2j], 2), ch.dp_frame_geometry(600, [1.0, 0.2j], 2)
    assert g1 is g2 and ch.dp_frame_geometry(601, [1.0, 0.2j], 2)["N_conv"] == g1["N_conv"] + 1
