"""processing()-level checks on the GPU: the drop-in call surface end to end (seeded channel -> HIP training loop ->
batched epilogue) against trajectories captured from the reference's own processing() with the same seeds.

Free-running trajectories are chaotic (the reference at 1 vs 4 CPU threads differs by 2.6e-2 in W after 100
steps), so beyond the first frames the comparison is statistical: converged SER within Monte-Carlo error, convergence
at a similar frame."""
import os
import sys

import numpy as np
import pytest
import scipy.io as io
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
PHI = np.array([0.0314, 0.0314], dtype=np.complex64)
TAU_PMD = 0.1e-12 * np.sqrt(1000)


def test_vaele_processing_vs_reference_trajectory():
    from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing
    g = load_golden("G7_runs")
    F, N = int(g["vaele_num_frames"]), int(g["vaele_N_frame_max"])
    SER, Var_est, var = processing("64-QAM", 2, 23, 0.0, 25, float(g["vaele_theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0",
                                   90e9, -26e-24, TAU_PMD, PHI, 170, seed=int(g["vaele_seed"]), verbose=False)
    assert SER.shape == (4, F) and Var_est.shape == (2, F) and var.shape == (2,) and SER.dtype == torch.float32 and not SER.is_cuda
    assert np.allclose(var.numpy(), g["vaele_var"])
    ref, ours = g["vaele_SER"], SER.numpy()
    # identical input frames (seeded generator) -> the first frames agree closely before chaos sets in
    assert np.max(np.abs(Var_est.numpy()[:, :3] - g["vaele_Var_est"][:, :3]) / g["vaele_Var_est"][:, :3]) < 1e-3
    assert np.max(np.abs(ours[:, :3] - ref[:, :3])) < 0.02
    # WHEN a run escapes the initial plateau is chaotic (rounding-level changes move it by tens of frames) and is not asserted on one seed:
    # tests/test_ensemble_gpu.py compares the escape-frame DISTRIBUTION of the HIP path with the oracle's on the same frame sets and places this
    # capture of the reference inside it.  Here: both lock, and the converged levels agree on the reference's own frames
    conv = lambda s: int(np.argmax((s < 0.1).all(0)))
    assert (ref[:, -1] < 0.1).all() and (ours[:, -1] < 0.1).all(), (conv(ours), conv(ref))
    # converged regime: mean SER over the frames after BOTH have converged (4 rows x >= 5 frames x ~870 symbols)
    lo = max(conv(ours), conv(ref)) + 4
    assert F - lo >= 5, (conv(ours), conv(ref))
    assert np.all(np.abs(ours[:, lo:].mean(1) - ref[:, lo:].mean(1)) < 6e-3), (ours[:, lo:].mean(1), ref[:, lo:].mean(1))
    assert ours[:, lo:].mean() < 0.04
    assert np.max(np.abs(Var_est.numpy()[:, lo:].mean(1) - g["vaele_Var_est"][:, lo:].mean(1)) / g["vaele_Var_est"][:, lo:].mean(1)) < 0.05


def test_vaeflex_processing_vs_reference_trajectory():
    from vae_equalizer_amd.func_VAEflex_DP_MQAM_shaping import processing
    g = load_golden("G7_runs")
    F, N = int(g["flex_num_frames"]), int(g["flex_N_frame_max"])
    SER, Var_est, var = processing("64-QAM", 2, 23, 0.0, 25, float(g["flex_theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0",
                                   90e9, -26e-24, TAU_PMD, PHI, 170, seed=int(g["flex_seed"]), verbose=False)
    assert SER.shape == (4, F)
    assert np.max(np.abs(Var_est.numpy()[:, 0] - g["flex_Var_est"][:, 0]) / g["flex_Var_est"][:, 0]) < 1e-4     # 30 steps in
    assert np.max(np.abs(Var_est.numpy()[:, 1] - g["flex_Var_est"][:, 1]) / g["flex_Var_est"][:, 1]) < 1e-3     # 60 steps in
    assert np.max(np.abs(Var_est.numpy() - g["flex_Var_est"]) / g["flex_Var_est"]) < 0.2        # chaotic after ~100 steps
    assert np.max(np.abs(SER.numpy()[:, 0] - g["flex_SER"][:, 0])) < 0.02
    assert np.all(np.abs(SER.numpy().mean(1) - g["flex_SER"].mean(1)) < 0.03)


def test_awgn_processing_vs_reference_trajectory():
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import processing
    g = load_golden("G7_runs")
    SER = processing("16-QAM", 2, 24, 0.0, 25, 5e-3, 350, 15000, 1200, 20, 2, "h1", seed=72, verbose=False)
    assert SER.shape == (10,) and SER.dtype == torch.float32
    assert abs(float(SER[0]) - float(g["awgn_SER"][0])) < 0.01            # 3 steps in: same data, same taps up to the coin flip
    assert np.max(np.abs(SER.numpy() - g["awgn_SER"])) < 0.05


def test_batch_equals_single_runs_bitwise():
    """Runs are independent workgroups: a run inside a batch of 5 == the same run alone."""
    from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
    runs = [DPRun(20 + i, [0.0, 0.0270955][i % 2], 0.01, 0.3, 2e-3 + 5e-4 * i, 90e9, seed=100 + i) for i in range(5)]
    kw = dict(mod="64-QAM", sps=2, M_est=25, batch_len=100, N_frame_max=500, num_frames=3, flex_step=10, channel="h0", tau_cd=-26e-24,
              tau_pmd=TAU_PMD, phiIQ=PHI, N_lrhalf=2)
    b = run_dp_batch(runs, **kw)
    for i in (0, 3):
        s = run_dp_batch([runs[i]], **kw)
        assert torch.equal(s["SER"][0], b["SER"][i]) and torch.equal(s["Var_est"][0], b["Var_est"][i])
        assert torch.equal(s["engine"].W[0], b["engine"].W[i])
    assert np.all(np.diff(b["var"][:, 0].numpy()[::2]) < 0)     # var follows SNR (shared_funcs.py:581)


def test_overlapped_frames_equal_serial_frames_bitwise(monkeypatch):
    """Below the resident-run count run_dp_batch puts the three stages of a frame on three streams (channel model of frame f + 1 and epilogue of frame
    f - 1 beside the training launch of frame f): every result bit-identical to the serial order (func_VAELE_DP_MQAM_shaping.py:43-89 is the serial loop)."""
    from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
    runs = [DPRun(20 + 2 * (i % 5), [0.0, 0.0270955][i % 2], 0.06 * np.pi, np.pi / 10, 2e-3 + 5e-4 * (i % 3), 90e9) for i in range(40)]
    runs[0].seed = 1234                                                         # the Philox key of the device generator
    kw = dict(mod="64-QAM", sps=2, M_est=25, batch_len=100, N_frame_max=3000, num_frames=12, flex_step=10, channel="h0", tau_cd=-26e-24,
              tau_pmd=TAU_PMD, phiIQ=PHI, N_lrhalf=5, generator="hip")
    a = run_dp_batch(runs, **kw)
    monkeypatch.setenv("VAEQ_SERIAL_FRAMES", "1")
    b = run_dp_batch(runs, **kw)
    for k in ("SER", "Var_est"):
        assert torch.equal(a[k], b[k]), k
    for k in ("W", "h", "mW", "vW", "mh", "vh", "step"):
        assert torch.equal(getattr(a["engine"], k), getattr(b["engine"], k)), k
    assert torch.isfinite(a["SER"]).all() and (a["SER"][:, :, 0] > 0.5).all()
    kw.update(flex=True, N_frame_max=1000, num_frames=4)                         # VAEflex windows through the same pipeline
    monkeypatch.delenv("VAEQ_SERIAL_FRAMES")
    c = run_dp_batch(runs[:6], **kw)
    monkeypatch.setenv("VAEQ_SERIAL_FRAMES", "1")
    d = run_dp_batch(runs[:6], **kw)
    assert torch.equal(c["SER"], d["SER"]) and torch.equal(c["Var_est"], d["Var_est"])


def test_torch_generator_path_trains():
    """On-device channel generator (row f1) feeding the kernel: loss decreases, outputs well-formed."""
    from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
    runs = [DPRun(23, 0.0, 0.0, 0.3, 2.5e-3, 90e9, seed=7) for _ in range(4)]
    r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 2000, 4, 10, "h0", -26e-24, TAU_PMD, PHI, 170, generator="torch")
    assert torch.isfinite(r["SER"]).all() and torch.isfinite(r["Var_est"]).all()
    assert (r["Var_est"][:, :, -1] < r["Var_est"][:, :, 0]).all()


def test_eval_run_dp_script_mat_schema(tmp_path, monkeypatch):
    """Eval_run_DP.main() on a tiny sweep: result tensor shapes and the .mat schema of the reference (:52-54, :99-114)."""
    from vae_equalizer_amd import Eval_run_DP as ev
    monkeypatch.setattr(ev, "SNR_vec", [20, 24]); monkeypatch.setattr(ev, "lr_optim_vec", [2.5e-3, 2e-3]); monkeypatch.setattr(ev, "iter", 2)
    monkeypatch.setattr(ev, "num_frames", 2); monkeypatch.setattr(ev, "N_frame_max", 400); monkeypatch.setattr(ev, "savePATH", str(tmp_path) + "/")
    monkeypatch.setattr(ev, "base_seed", 5)
    name, d = ev.main()
    assert d["SER"].shape == (4, 2, 1, 1, 1, 1, 2, 1, 1, 1, 2, 2) and d["Var_est"].shape == (2, 2, 1, 1, 1, 1, 2, 1, 1, 1, 2, 2)
    assert d["var_real"].shape == (2, 2, 1, 1, 1, 1, 2, 1, 1, 1, 2, 1)
    m = io.loadmat(name)["dict"]
    assert set(m.dtype.names) == {"SER", "Var_est", "var_real", "SNR", "nu", "theta_diff", "theta", "M", "lr", "batch_len", "symb_rate", "symb_step"}
    assert "SERvsSNR_VAE_DP_64-QAM_N_lrhalf_170_N_train_400_" in name
    assert np.isfinite(d["SER"]).all() and (d["var_real"][0, 0] > d["var_real"][0, 1]).all()


def test_eval_run_awgn_script_mat_schema(tmp_path, monkeypatch):
    from vae_equalizer_amd import Eval_run_shaping_vaele as ev
    monkeypatch.setattr(ev, "iter", 2); monkeypatch.setattr(ev, "num_epochs", 4); monkeypatch.setattr(ev, "N_valid", 2000)
    monkeypatch.setattr(ev, "savePATH", str(tmp_path) + "/"); monkeypatch.setattr(ev, "base_seed", 3)
    name, d = ev.main()
    assert d["SER"].shape == (1, 1, 1, 1, 1, 1, 2, 2)
    m = io.loadmat(name)["dict"]
    assert set(m.dtype.names) == {"SER", "SNR", "M", "lr", "N_train", "nu"}
    assert "SERvsSNR_VAELE_shaping_0_h1_64-QAM_2_2000_2_1200_" in name


@pytest.mark.parametrize("net_type,gen", [("Net", "hip"), ("Net_BN", "numpy")])
def test_eval_run_vaenn_script_mat_schema(tmp_path, monkeypatch, net_type, gen):
    """Eval_run_vaenn.main() on a tiny sweep: result tensor shape and the .mat schema of the reference (:36, :58-68)."""
    from vae_equalizer_amd import Eval_run_vaenn as ev
    monkeypatch.setattr(ev, "iter", 2); monkeypatch.setattr(ev, "num_epochs", 4); monkeypatch.setattr(ev, "N_valid", 2000)
    monkeypatch.setattr(ev, "train_len", 900); monkeypatch.setattr(ev, "SNR_vec", [20, 24]); monkeypatch.setattr(ev, "net_type_vec", [net_type])
    monkeypatch.setattr(ev, "savePATH", str(tmp_path) + "/"); monkeypatch.setattr(ev, "base_seed", 3); monkeypatch.setattr(ev, "generator", gen)
    name, d = ev.main()
    assert d["SER"].shape == (2, 1, 1, 1, 1, 1, 2, 2) and np.isfinite(d["SER"]).all() and (d["SER"] > 0.3).all()       # 4 epochs: far from locked
    m = io.loadmat(name)["dict"]
    assert set(m.dtype.names) == {"SER", "SNR", "k2", "k1", "M", "lr", "N_train"}
    assert f"SERvsSNR_{net_type}_h1_64-QAM_2_2000_2_900_" in name


def test_hip_epilogue_on_converged_reference_frame():
    """vaeq_dp_epilogue on G5's converged frame == the reference's own shift / swap / SER results."""
    from vae_equalizer_amd.engine import dp_epilogue
    g = load_golden("G5_dp_epilogue")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    r = dp_epilogue(t(g["out_train"])[None], t(g["out_const"])[None], t(g["data"])[None], g["amp_levels"], float(g["nu_sc"]), g["var"],
                    batch_len=int(g["B"]))
    assert np.array_equal(r["shift_q"][0].cpu().numpy(), g["shifts"][-1, 0]) and int(r["r_q"][0]) == g["rs"][-1, 0]
    assert np.array_equal(r["shift_c"][0].cpu().numpy(), g["shifts"][-1, 1]) and int(r["r_c"][0]) == g["rs"][-1, 1]
    assert np.allclose(r["SER"][0].cpu().numpy(), g["SER_valid"][:, -1], atol=1e-6)


@pytest.mark.parametrize("batch_len", [None, 100])
@pytest.mark.parametrize("n", [2, 8])
def test_hip_epilogue_random_batch_vs_oracle(batch_len, n):
    """R=12 synthetic runs (delays, swaps, rotations, IQ flips, different noise / shaping): HIP epilogue == numpy oracle per run."""
    import oracle
    from vae_equalizer_amd.engine import dp_epilogue
    lev_all = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = (lev_all / np.sqrt(np.mean(lev_all ** 2) * 2)).astype(np.float32)
    rng = np.random.default_rng(21 + n)
    R, N = 12, 1000
    qs, ys, ds, nus, vars_ = [], [], [], [], []
    for i in range(R):
        lev = rng.integers(0, n, (2, 2, N))
        clean = amp[lev].astype(np.float32)
        rot = [clean, np.stack([-clean[:, 1], clean[:, 0]], 1), -clean, np.stack([clean[:, 1], -clean[:, 0]], 1)][i % 4]
        if i % 5 == 4:
            rot = np.stack([rot[:, 0], -rot[:, 1]], 1)
        y = rot + (0.02 + 0.03 * i) * rng.standard_normal(rot.shape).astype(np.float32)
        sw, d = i % 2, int(rng.integers(-9, 10))
        y = np.roll(y, sw, axis=0)
        dl = (d, d) if sw else (d, int(rng.integers(-9, 10)))
        y = np.stack([np.roll(y[0], dl[0], -1), np.roll(y[1], dl[1], -1)])
        nu_sc = float(rng.uniform(0, 1.2)); v = rng.uniform(0.002, 0.02, 2).astype(np.float32)
        qs.append(oracle.dp_soft_dec(y, v, amp, nu_sc)); ys.append(y); ds.append(amp[lev].astype(np.float16)); nus.append(nu_sc); vars_.append(v)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a))).cuda()
    r = dp_epilogue(t(qs), t(ys), t(ds), amp, torch.tensor(nus, dtype=torch.float32), t(vars_), batch_len)
    for i in range(R):
        o = oracle.dp_frame_epilogue(qs[i], ys[i], ds[i], amp, nus[i], vars_[i], batch_len=batch_len)
        assert np.array_equal(r["shift_q"][i].cpu().numpy(), o["shift_q"]) and int(r["r_q"][i]) == o["r_q"], i
        assert np.array_equal(r["shift_c"][i].cpu().numpy(), o["shift_c"]) and int(r["r_c"][i]) == o["r_c"], i
        assert np.allclose(r["SER"][i].cpu().numpy(), o["SER"], atol=2.5e-3), (i, r["SER"][i], o["SER"])     # <= 2 symbols of ~900 at a threshold


@pytest.mark.parametrize("N,batch_len,n,offgrid", [(1000, 100, 8, False), (10000, 100, 8, False), (9900, None, 8, False), (1002, None, 4, False), (3000, 50, 2, False),
                                                 (2000, 100, 8, True)])
def test_lds_resident_epilogue_equals_rereading_epilogue(N, batch_len, n, offgrid, monkeypatch):
    """vaeq_dp_epilogue_compact's kernels -- dp_epilogue_compact_kernel (both correlations per staged TX tile, branch-free walks; with and without the
    frame's TX levels resident in LDS, vaeq_epilogue_lds.h) vs the re-reading dp_epilogue_kernel (VAEQ_EPI_REREAD=1) -- agree bit for bit on synthetic runs (delays, swaps, rotations, IQ flips), also for frame lengths
    that are no multiple of four and for a TX reference that is NOT the fp16 image of the levels (the radius walk then reads TX itself); and with the
    numpy oracle on shifts / swaps exactly, SER within two symbols at a decision threshold."""
    import oracle
    from vae_equalizer_amd.engine import dp_epilogue_compact
    lev_all = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = (lev_all / np.sqrt(np.mean(lev_all ** 2) * 2)).astype(np.float32)
    rng = np.random.default_rng(5 + N + n)
    R = 9
    qs, ys, ds, nus, vars_ = [], [], [], [], []
    for i in range(R):
        lev = rng.integers(0, n, (2, 2, N))
        clean = amp[lev].astype(np.float32)
        rot = [clean, np.stack([-clean[:, 1], clean[:, 0]], 1), -clean, np.stack([clean[:, 1], -clean[:, 0]], 1)][i % 4]
        if i % 5 == 4:
            rot = np.stack([rot[:, 0], -rot[:, 1]], 1)
        y = (0.8 + 0.05 * i) * rot + (0.02 + 0.03 * i) * rng.standard_normal(rot.shape).astype(np.float32)
        sw, d = i % 2, int(rng.integers(-9, 10))
        y = np.roll(y, sw, axis=0)
        dl = (d, d) if sw else (d, int(rng.integers(-9, 10)))
        y = np.stack([np.roll(y[0], dl[0], -1), np.roll(y[1], dl[1], -1)]).astype(np.float32)
        nu_sc = float(rng.uniform(0, 1.2)); v = rng.uniform(0.002, 0.02, 2).astype(np.float32)
        tx = amp[lev].astype(np.float16)
        if offgrid:
            tx = (tx.astype(np.float32) * (1 + 0.01 * rng.standard_normal(tx.shape))).astype(np.float16)   # same levels under rint(scale t + scale), other radii
        qs.append(oracle.dp_soft_dec(y, v, amp, nu_sc)); ys.append(y); ds.append(tx); nus.append(nu_sc); vars_.append(v)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a))).cuda()
    q, yt, dt, amp_t = t(qs), t(ys), t(ds), torch.tensor(amp, device="cuda")
    eq = torch.einsum("i,rpin->rpn", amp_t, q[:, :, :n]).contiguous()
    dec = torch.stack([q[:, 0, :n].argmax(1), q[:, 0, n:].argmax(1), q[:, 1, :n].argmax(1), q[:, 1, n:].argmax(1)], 1).reshape(R, 2, 2, N).to(torch.int8)
    nu_t, var_t = torch.tensor(nus, dtype=torch.float32, device="cuda"), t(vars_)
    new = dp_epilogue_compact(eq, dec, yt, dt, amp_t, nu_t, var_t, batch_len)
    monkeypatch.setenv("VAEQ_EPI_NOTXC", "1")                                    # the same kernel without the TX level cache in LDS
    mid = dp_epilogue_compact(eq, dec, yt, dt, amp_t, nu_t, var_t, batch_len)
    monkeypatch.setenv("VAEQ_EPI_REREAD", "1")
    old = dp_epilogue_compact(eq, dec, yt, dt, amp_t, nu_t, var_t, batch_len)
    for k in old:
        assert torch.equal(old[k], new[k]) and torch.equal(old[k], mid[k]), (k, old[k], new[k], mid[k])
    eq_h, dec_h = eq.cpu().numpy(), dec.cpu().numpy()
    for i in range(R):
        if offgrid:
            continue                                             # (the oracle is pinned on on-grid references; the two kernels agreeing is the check here)
        o = oracle.dp_frame_epilogue(qs[i], ys[i], ds[i], amp, nus[i], vars_[i], batch_len=batch_len)
        assert np.array_equal(new["shift_c"][i].cpu().numpy(), o["shift_c"]) and int(new["r_c"][i]) == o["r_c"], i
        assert np.allclose(new["SER"][i, :2].cpu().numpy(), o["SER"][:2], atol=2.5e-3), (i, new["SER"][i], o["SER"])
        # the soft-demapper rows come from eq / dec as torch derives them from q (argmax ties, fma order): compared on the shift only when unambiguous
        if np.array_equal(new["shift_q"][i].cpu().numpy(), o["shift_q"]) and int(new["r_q"][i]) == o["r_q"]:
            assert np.allclose(new["SER"][i, 2:].cpu().numpy(), o["SER"][2:], atol=2.5e-3), (i, new["SER"][i], o["SER"])


def test_awgn_batch_equals_single_runs():
    """AWGN runs batched in one launch per epoch == the same runs alone (independent workgroups), incl. different SNR / shaping."""
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import run_awgn_batch
    runs = [dict(SNR=20 + 2 * i, nu=[0.0, 0.0270955][i % 2], lr_optim=5e-3, seed=300 + i) for i in range(4)]
    kw = dict(mod="64-QAM", sps=2, M_est=25, batch_len=350, N_valid=3000, N_train=1200, num_epochs=4, epe=2, channel="h1")
    b = run_awgn_batch(runs, **kw)
    assert b.shape == (4, 2) and torch.isfinite(b).all()
    for i in (1, 2):
        assert torch.equal(run_awgn_batch([runs[i]], **kw)[0], b[i])


def test_full_config3_run_vs_reference_statistics():
    """SURVEY config 3 at full size (170 frames x 10 000 symbols = 17 000 Adam steps, Eval_run_DP.py defaults, lr 2.5e-3) with the
    frames the reference saw under seed 1234 (tools/capture_golden.py --full-run: 604 s on the CPU there): SER curve and noise
    estimate agree within Monte-Carlo error."""
    from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing
    g = load_golden("G7_full_runs")
    SER, Var_est, var = processing("64-QAM", 2, 23, 0.0, 25, 0.06 * np.pi, np.pi / 10, 2.5e-3, 100, 10000, 170, 10, "h0", 90e9, -26e-24,
                                   TAU_PMD, PHI, 170, seed=1234, verbose=False)
    ref, ours = g["full_SER"], SER.numpy()
    assert np.allclose(var.numpy(), g["full_var"])
    conv = lambda s: int(np.argmax((s < 0.1).all(0)))
    assert abs(conv(ours) - conv(ref)) <= 4, (conv(ours), conv(ref))                     # reference: frame 16
    # same noise realisations on both sides -> the converged SERs agree far inside the per-frame MC sigma (1.8e-3)
    assert np.all(np.abs(ours[:, -30:].mean(1) - ref[:, -30:].mean(1)) < 1.5e-3), (ours[:, -30:].mean(1), ref[:, -30:].mean(1))
    assert np.max(np.abs(ours[:, 40:] - ref[:, 40:])) < 0.012
    ve_o, ve_r = Var_est.numpy()[:, -30:].mean(1), g["full_Var_est"][:, -30:].mean(1)
    assert np.max(np.abs(ve_o - ve_r) / ve_r) < 0.01, (ve_o, ve_r)
    snr_est = 10 * np.log10(1.0 / ve_o.mean())                                          # pow_mean (= 1 at nu = 0) / Var_est (func_VAELE_DP...:68)
    assert abs(snr_est - 21.97) < 0.15                                                  # the survey's measured reference value


def test_awgn_config2_run_vs_reference_statistics():
    """SURVEY config 2 (AWGN 64-QAM + PCS nu=0.0270955, h1, SNR 24 dB, 25 taps, 500 epochs x 3 minibatches of 350, validation on
    15 000 symbols every 2nd epoch) with the frames the reference saw under seed 73: SER curve within Monte-Carlo error."""
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import processing
    g = load_golden("G7_awgn_cfg2")
    SER = processing("64-QAM", 2, 24, 0.0270955, 25, 5e-3, 350, 15000, 1200, 500, 2, "h1", seed=int(g["seed"]), verbose=False).numpy()
    ref = g["SER"]
    assert SER.shape == ref.shape == (250,)
    assert abs(SER[0] - ref[0]) < 0.01                                        # first validation: 3 steps in
    # When the blind equalizer locks is chaotic: on these very frames the CPU oracle locks at validation 47-53 in fp32 and at
    # 57 / 132 / 143 in fp64 depending on the sign of the noise-driven first step of the scale tap (reference: 56).  What is
    # pinned is that it locks inside that band and where it ends up.
    conv = lambda s: int(np.argmax(s < 0.01))
    assert 30 <= conv(SER) <= 200, conv(SER)
    assert abs(SER[-50:].mean() - ref[-50:].mean()) < 3e-4, (SER[-50:].mean(), ref[-50:].mean())   # ~1.1e-3 both
    assert np.max(np.abs(SER[-40:] - ref[-40:])) < 1.5e-3


def test_eval_run_dp_sharded_two_ranks(tmp_path):
    """The sweep script under torch.distributed.run with 2 ranks (gloo, both on this GPU): run r -> rank r % 2, one all_gather, rank 0
    writes the .mat -- identical to the single-process result (runs are independent and seeded per sweep point)."""
    import glob
    import subprocess
    import sys
    from conftest import ROOT
    from vae_equalizer_amd import Eval_run_DP as ev
    import os
    env = dict(os.environ, VAEQ_DIST_BACKEND="gloo", VAEQ_SINGLE_DEVICE="1", MPLBACKEND="Agg")
    d2, d1 = str(tmp_path / "two") + "/", str(tmp_path / "one") + "/"
    os.makedirs(d2); os.makedirs(d1)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", os.path.join(ROOT, "tests", "_run_eval_dp_small.py"), d2], check=True, env=env, timeout=600,
                   cwd=ROOT)
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_run_eval_dp_small.py"), d1], check=True, timeout=600, cwd=ROOT)
    m2 = io.loadmat(glob.glob(d2 + "*.mat")[0])["dict"]
    m1 = io.loadmat(glob.glob(d1 + "*.mat")[0])["dict"]
    for k in ("SER", "Var_est", "var_real"):
        assert np.array_equal(m2[k][0, 0], m1[k][0, 0]), k
    assert m2["SER"][0, 0].shape == (4, 2, 1, 1, 1, 1, 2, 1, 1, 1, 2, 2)


def test_eval_run_dp_untouched_defaults_are_fast(tmp_path):
    """The drop-in script with its constants untouched (Eval_run_DP.py:18-49: 3 learning rates x iter 5 = 15 unseeded runs x 170 frames x 10 000
    symbols; the reference needs 15 x ~600 s of CPU for it) runs on the device end to end -- generator included -- in seconds: the default
    generator for unseeded sweeps is the on-device simulator (dp_runs.resolve_generator), no switch needed."""
    import json
    import subprocess
    import scipy.io as io
    d = str(tmp_path) + "/"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_run_eval_dp_defaults.py"), d], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert info["generator"] is None and info["base_seed"] is None
    assert info["seconds"] < 10.0, info                                          # main() wall time incl. library load, tables, 170 frames, .mat
    m = io.loadmat(info["mat"])["dict"]
    SER = m["SER"][0, 0]
    assert SER.shape == (4, 1, 1, 1, 1, 1, 3, 1, 1, 1, 5, 170)
    tail = SER[..., -30:].reshape(4, 15, 30)
    locked = (tail < 0.1).all(axis=(0, 2))
    assert locked.sum() >= 13, tail.mean(-1)                                     # (nearly) every run locks; ~1 in 1000 ends in the polarisation singularity
    assert abs(tail[:, locked].mean() - 0.0314) < 4e-3, tail[:, locked].mean()  # the reference's converged SER at 23 dB (G7_full_runs)


def test_eval_run_shaping_vaele_untouched_defaults_are_fast(tmp_path):
    """The AWGN drop-in script with its constants untouched (Eval_run_shaping_vaele.py:9-22: iter 20 unseeded runs x 500 epochs, a validation on
    15 000 fresh symbols every second epoch) on the device end to end, validation frames generated clean and noised while they are read
    (vaeq_gen_awgn_clean -> vaeq_awgn_validate_gen): seconds, every run locks."""
    import json
    import subprocess
    import scipy.io as io
    d = str(tmp_path) + "/"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_run_eval_awgn_defaults.py"), d], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert info["generator"] is None and info["base_seed"] is None
    assert info["seconds"] < 10.0, info
    SER = io.loadmat(info["mat"])["dict"]["SER"][0, 0]
    assert SER.shape == (1, 1, 1, 1, 1, 1, 20, 250)
    tail = SER.reshape(20, 250)[:, -50:]
    lvl = tail.mean(axis=1)
    # uniform 64-QAM at 24 dB over h1: a run either locks (SER 0.0088 ... 0.0100 over the last 50 validations) or is still on the blind
    # equaliser's initial plateau (0.867) after 500 epochs -- about one run in ten, different runs every sweep, the same in the two-step form
    # (measured: 1, 3, 1, 4 of 40, 1); the lock statistics themselves are tests/test_ensemble_gpu.py's subject (HIP vs oracle)
    # a run that locks INSIDE (or shortly before) the 50-validation tail has a tail mean anywhere between the two levels -- unseeded sweep, fresh entropy every
    # time: such a run is neither "locked" nor "stuck" for the level assertions (this test failed once in ~25 full-suite runs on exactly that)
    S2 = SER.reshape(20, 250)
    first = np.where((S2 < 0.1).any(axis=1), (S2 < 0.1).argmax(axis=1), 250)     # first validation below 0.1
    locked, stuck = first < 160, first == 250                                   # locked with >= 40 validations (80 epochs) to settle before the tail
    assert locked.sum() >= 12 and (~locked & ~stuck).sum() <= 4, (first, lvl)   # P(more than 8 of 20 not locked by then) < 1e-3
    assert lvl[locked].max() < 1.25 * np.median(lvl[locked]) and 0.004 < np.median(lvl[locked]) < 0.015, (first, lvl)
    assert (np.abs(lvl[stuck] - 0.867) < 0.02).all(), (first, lvl)
    assert SER.reshape(20, 250)[:, 0].min() > 0.3                                # and starts unconverged: the curve is a training curve


def test_awgn_config2_device_pipeline_monte_carlo():
    """Config 2 end to end on the device -- vaeq_gen_awgn -> vaeq_awgn_train (wave kernel) -> vaeq_awgn_validate -- for 24 independent
    runs: every run locks, and the converged SER agrees with the reference's curve (G7_awgn_cfg2, ~1.1e-3) within Monte-Carlo error."""
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import run_awgn_batch
    g = load_golden("G7_awgn_cfg2")
    runs = [dict(SNR=24, nu=0.0270955, lr_optim=5e-3, seed=None) for _ in range(24)]
    SER = run_awgn_batch(runs, "64-QAM", 2, 25, 350, 15000, 1200, 500, 2, "h1", generator="hip", seed=99).numpy()
    assert SER.shape == (24, 250)
    conv = np.array([int(np.argmax(s < 0.01)) for s in SER])
    locked = (SER[:, -50:] < 0.01).all(1)                                      # locked before validation 200
    # when the blind equalizer locks is chaotic (see test_awgn_config2_run_vs_reference_statistics: 47 ... 143 on ONE set of frames
    # depending on rounding); nearly all of 24 independent runs lock before validation 200, every one before the end
    assert locked.sum() >= 20 and (SER[:, -5:] < 0.01).all(), conv
    assert np.all(conv >= 20) and np.median(conv) < 160, conv                 # neither instantly nor never (reference: 56)
    tail = SER[locked, -50:].mean()                                           # >= 20 x 50 x 15000 symbols
    assert abs(tail - g["SER"][-50:].mean()) < 1.5e-4, (tail, g["SER"][-50:].mean())


def test_vaenn_processing_vs_reference_trajectory():
    """Row f3 through the drop-in call surface: func_VAENN_MQAM.processing with the frames and the initial network the reference saw
    under seed 84 (16-QAM, SNR 20 dB, 120 epochs x 4 minibatches of 300, validation on 5000 symbols every 2nd epoch)."""
    from vae_equalizer_amd.func_VAENN_MQAM import processing
    g = load_golden("G8_vaenn_run")
    SER = processing("16-QAM", 2, 20, 25, 25, 3, 4e-3, 300, 5000, 1200, 120, 2, "h1", "Net", seed=int(g["seed"]), theta0=g["theta0"],
                     generator="numpy", verbose=False).numpy()
    ref = g["SER"]
    assert SER.shape == ref.shape == (60,)
    assert np.max(np.abs(SER[:5] - ref[:5])) < 0.02                            # same frames, same start: 20 steps in
    conv = lambda s: int(np.argmax(s < 0.02))
    assert abs(conv(SER) - conv(ref)) <= 8, (conv(SER), conv(ref))             # reference: validation 25
    assert abs(SER[-20:].mean() - ref[-20:].mean()) < 8e-4, (SER[-20:].mean(), ref[-20:].mean())
    with pytest.raises(UnboundLocalError):
        processing("16-QAM", 2, 20, 25, 25, 3, 4e-3, 300, 5000, 1200, 2, 2, "h1", "Net_LN", verbose=False)


def test_vaennbn_processing_vs_reference_trajectory():
    """Net_BN through processing(): the frames and the initial network the reference saw under seed 114 (120 epochs)."""
    from vae_equalizer_amd.func_VAENN_MQAM import processing
    g = load_golden("G11_vaennbn_run")
    SER = processing("16-QAM", 2, 20, 25, 25, 3, 4e-3, 300, 5000, 1200, 120, 2, "h1", "Net_BN", seed=int(g["seed"]), theta0=g["theta0"],
                     generator="numpy", verbose=False).numpy()
    ref = g["SER"]
    assert SER.shape == ref.shape == (60,)
    assert np.max(np.abs(SER[:5] - ref[:5])) < 0.03
    conv = lambda s: int(np.argmax(s < 0.02))
    assert abs(conv(SER) - conv(ref)) <= 10, (conv(SER), conv(ref))            # reference: validation 37
    assert abs(SER[-15:].mean() - ref[-15:].mean()) < 8e-4, (SER[-15:].mean(), ref[-15:].mean())


def test_vaenn_device_pipeline_monte_carlo():
    """vaeq_gen_awgn (fixed noise level) -> vaeq_nn_train -> vaeq_nn_validate for 16 independently initialised runs: all lock, and the
    converged SER agrees with the reference's curve within Monte-Carlo error."""
    from vae_equalizer_amd.func_VAENN_MQAM import run_vaenn_batch
    g = load_golden("G8_vaenn_run")
    runs = [dict(SNR=20, lr_optim=4e-3, seed=None) for _ in range(16)]
    SER = run_vaenn_batch(runs, "16-QAM", 2, 25, 25, 3, 300, 5000, 1200, 120, 2, "h1", generator="hip", seed=5).numpy()
    assert SER.shape == (16, 60)
    tail = SER[:, -10:].mean(1)
    locked = tail < 0.01
    # how fast the blind CNN equalizer escapes its start depends on the random initialisation: most runs lock within the 120 epochs
    # (the reference's own run: validation 25 of 60), the rest are still descending
    assert locked.sum() >= 10, tail
    assert (SER[~locked, -1] < SER[~locked, 0] - 0.05).all(), SER[~locked][:, [0, -1]]
    assert abs(SER[locked, -20:].mean() - g["SER"][-20:].mean()) < 6e-4, (SER[locked, -20:].mean(), g["SER"][-20:].mean())


def test_vaeflex_converging_run_vs_reference():
    """Config 4 (optical DP 64-QAM VAEflex, window 100 / step 10) through processing() on the 70 frames x 2000 symbols the reference saw
    under seed 91 (13 300 window steps): same convergence behaviour, converged SER and noise estimate within Monte-Carlo error."""
    from vae_equalizer_amd.func_VAEflex_DP_MQAM_shaping import processing
    g = load_golden("G9_flex_run")
    F, N = int(g["num_frames"]), int(g["N_frame_max"])
    SER, Var_est, var = processing("64-QAM", 2, 23, 0.0, 25, float(g["theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0", 90e9, -26e-24,
                                   TAU_PMD, PHI, 170, seed=int(g["seed"]), verbose=False)
    ours, ref = SER.numpy(), g["SER"]
    assert ours.shape == ref.shape == (4, F)
    assert np.max(np.abs(ours[:, :2] - ref[:, :2])) < 0.03                     # same frames: the first 380 steps agree closely
    conv = lambda s: int(np.argmax((s < 0.1).all(0)))
    assert (ref[:, -1] < 0.1).all() and (ours[:, -1] < 0.1).all()
    assert abs(conv(ours) - conv(ref)) <= 12, (conv(ours), conv(ref))
    lo = max(conv(ours), conv(ref)) + 4
    assert F - lo >= 8, (conv(ours), conv(ref))
    assert np.all(np.abs(ours[:, lo:].mean(1) - ref[:, lo:].mean(1)) < 6e-3), (ours[:, lo:].mean(1), ref[:, lo:].mean(1))
    ve, vr = Var_est.numpy()[:, lo:].mean(1), g["Var_est"][:, lo:].mean(1)
    assert np.max(np.abs(ve - vr) / vr) < 0.05, (ve, vr)


def test_vaele_pcs_run_vs_reference():
    """Config 5's shape: optical DP 64-QAM with probabilistic shaping (nu = 0.0270955, H = 5.72 bit) through processing() on the
    200 frames x 3000 symbols the reference saw under seed 101 (6000 minibatch steps): convergence frame, converged SER (both
    estimators: the constellation-based one uses the PCS-aware decision thresholds) and noise estimate."""
    from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing
    g = load_golden("G10_pcs_run")
    F, N = int(g["num_frames"]), int(g["N_frame_max"])
    SER, Var_est, var = processing("64-QAM", 2, 23, float(g["nu"]), 25, float(g["theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0", 90e9,
                                   -26e-24, TAU_PMD, PHI, 170, seed=int(g["seed"]), verbose=False)
    ours, ref = SER.numpy(), g["SER"]
    assert np.allclose(var.numpy(), g["var"], rtol=1e-6)
    assert np.max(np.abs(ours[:, :2] - ref[:, :2])) < 0.03
    conv = lambda s: int(np.argmax((s < 0.1).all(0)))
    assert (ref[:, -1] < 0.1).all() and (ours[:, -1] < 0.1).all(), (conv(ours), conv(ref))   # both lock; WHEN: tests/test_ensemble_gpu.py (distribution vs the oracle)
    lo = max(conv(ours), conv(ref)) + 4
    assert F - lo >= 20, (conv(ours), conv(ref))
    assert np.all(np.abs(ours[:, lo:].mean(1) - ref[:, lo:].mean(1)) < 2.5e-3), (ours[:, lo:].mean(1), ref[:, lo:].mean(1))
    ve, vr = Var_est.numpy()[:, lo:].mean(1), g["Var_est"][:, lo:].mean(1)
    assert np.max(np.abs(ve - vr) / vr) < 0.05, (ve, vr)


# ------------------------------------------------------------------ config 5: optical DP 64-QAM + PCS, SNR x shaping x seed sweep
@pytest.mark.parametrize("name", ["G13_cfg5_nu0872_snr20", "G13_cfg5_nu1222_snr28"])
def test_config5_heavy_shaping_runs_vs_reference(name):
    """Two on-grid points of config 5's sweep vectors (Eval_run_DP.py:24,34) through processing() on the 200 frames x 3000 symbols the reference
    saw.  At these shaping strengths (H = 4.6 / 4.125 bit) the reference's blind equaliser does NOT lock within the run -- all four SER estimates
    stay at ~0.85-0.9 and the noise estimate at 5-6 x (nu .0872, 20 dB) resp. 35 x (nu .1222, 28 dB) the true variance -- and neither must this
    implementation: same plateau, frame window by frame window."""
    from vae_equalizer_amd.func_VAELE_DP_MQAM_shaping import processing
    g = load_golden(name)
    F, N = int(g["num_frames"]), int(g["N_frame_max"])
    SER, Var_est, var = processing("64-QAM", 2, float(g["SNR"]), float(g["nu"]), 25, float(g["theta_diff"]), np.pi / 10, 2.5e-3, 100, N, F, 10, "h0", 90e9,
                                   -26e-24, TAU_PMD, PHI, 170, seed=int(g["seed"]), verbose=False)
    ours, ref, ve, vr = SER.numpy(), g["SER"], Var_est.numpy(), g["Var_est"]
    assert np.allclose(var.numpy(), g["var"], rtol=1e-6)
    # before chaos sets in: frame 0 (30 steps) agrees to 0.2 %, frame 1 to 2.5 %, frame 2 up to 6 % (the 28 dB point, var = 1.5e-4, is the touchiest)
    assert np.max(np.abs(ours[:, :2] - ref[:, :2])) < 0.05 and np.max(np.abs(ve[:, :1] - vr[:, :1]) / vr[:, :1]) < 0.02
    assert np.max(np.abs(ve[:, :2] - vr[:, :2]) / vr[:, :2]) < 0.05
    assert ref[:, 20:].min() > 0.6 and ours[:, 20:].min() > 0.6                        # neither locks
    for a, b in ((0, 20), (20, 60), (60, 120), (120, 200)):
        # an unlocked equaliser wanders on its plateau, chaotically in the SER of a window of ONE run: the plateau's SER is compared as an ensemble
        # against the oracle in tests/test_ensemble_gpu.py::test_heavy_shaping_plateau_ensemble_vs_oracle; on this single run the noise estimate is
        # the robust statistic of the plateau
        assert np.max(np.abs(ve[:, a:b].mean(1) - vr[:, a:b].mean(1)) / vr[:, a:b].mean(1)) < 0.08, (a, b, ve[:, a:b].mean(1), vr[:, a:b].mean(1))
    assert ve[:, 100:].mean() > 4 * float(var[0])                                       # the plateau's noise estimate, far above the true variance


def test_config5_grid_one_batch_on_one_gpu(tmp_path, monkeypatch):
    """BASELINE config 5, script-faithful: nu in {0, .0270955, .0872449, .1222578} x SNR in {20..28} x 3 learning rates x iter = 5 = 300 runs of
    170 frames x 10 000 symbols through the drop-in sweep script in ONE batch on one GPU (Eval_run_DP.py:24,34,67-95; the N = 1 anchor of the
    8-way shard).  Checks the .mat schema, the reference's on-grid behaviour (four script-size captures, G13_cfg5_full_*: the two light shapings
    lock, the two heavy ones never do) and that SER falls with the SNR until it reaches the tracking floor (from ~24 dB on the residual errors are
    the equaliser's lag behind the 0.06 pi / frame polarisation drift, not noise: 0.0114 / 0.0110 / 0.0127 at 24 / 26 / 28 dB for H = 5.72 bit)."""
    import time
    from vae_equalizer_amd import Eval_run_DP as ev
    NU, SNR = [0, 0.0270955, 0.0872449, 0.1222578], [20, 22, 24, 26, 28]
    monkeypatch.setattr(ev, "nu_vec", NU); monkeypatch.setattr(ev, "SNR_vec", SNR); monkeypatch.setattr(ev, "generator", "hip")
    monkeypatch.setattr(ev, "base_seed", 5); monkeypatch.setattr(ev, "savePATH", str(tmp_path) + "/")
    assert (ev.iter, ev.num_frames, ev.N_frame_max, ev.lr_optim_vec, ev.loss_type, ev.mod) == (5, 170, 10000, [2.5e-3, 2e-3, 3e-3], "VAE", "64-QAM")
    t0 = time.time()
    name, d = ev.main()
    wall = time.time() - t0
    print(f"config 5: 300 runs x 170 frames x 10 000 symbols in {wall:.1f} s on one GPU (first call of the process: library / module loads included; a second call "
          "takes 0.4 s = 2.3 ms per frame, tools/probe_config5_profile.py)")
    m = io.loadmat(name)["dict"]
    assert m["SER"][0, 0].shape == (4, 5, 1, 4, 1, 1, 3, 1, 1, 1, 5, 170) and m["Var_est"][0, 0].shape == (2, 5, 1, 4, 1, 1, 3, 1, 1, 1, 5, 170)
    assert m["var_real"][0, 0].shape == (2, 5, 1, 4, 1, 1, 3, 1, 1, 1, 5, 1)
    assert np.allclose(np.ravel(m["nu"][0, 0]), NU) and np.allclose(np.ravel(m["SNR"][0, 0]), SNR)
    S, V = d["SER"], d["Var_est"]
    assert np.isfinite(S).all() and np.isfinite(V).all() and wall < 120
    tail = lambda A, s, n: A[:, s, 0, n, 0, 0, :, 0, 0, 0, :, -30:].mean(-1)             # [rows, lr, iter]: mean over the last 30 frames
    # light shaping locks; SER falls with the SNR down to the floor the 0.06 pi/frame drift leaves (reached at ~26 dB).  About one run in a thousand
    # ends in the blind equaliser's polarisation singularity instead (both outputs on one polarisation: one polarisation's rows stay at ~0.93) --
    # 1 of 1200 with either form of the frame generator, at different seeds (profiles/r02_final/lock_rate_probe.txt) -- so lock is asserted for all
    # but at most two of the 150 runs and the statistics are taken over the locked ones
    lock = {n: np.array([(tail(S, s, n) < 0.2).all(0) for s in range(5)]) for n in (0, 1)}               # [SNR, lr, iter]
    assert sum(int((~lock[n]).sum()) for n in (0, 1)) <= 2, {n: np.argwhere(~lock[n]) for n in (0, 1)}
    for n in (0, 1):
        ser = np.array([tail(S, s, n)[:, lock[n][s]].mean() for s in range(5)])
        assert ser[0] > 1.5 * ser[1] > 1.5 * 1.2 * ser[2] and (ser[3:] < 1.25 * ser[2]).all() and (ser[3:] > 0.3 * ser[2]).all(), ser
        assert np.array([tail(V, s, n)[:, lock[n][s]].mean() for s in range(5)]).argsort().tolist() == [4, 3, 2, 1, 0]      # noise estimate falls with the SNR
    assert tail(S, 0, 1)[:, lock[1][0]].mean() < 0.5 * tail(S, 0, 0)[:, lock[0][0]].mean()       # H = 5.72 bit needs fewer errors at equal SNR
    # heavy shaping: (nearly) no run locks within the 170 frames, at any SNR -- the escape from the plateau is chaotic, one run in 150 has been seen
    # to lock (SER 0.003) with one build of the kernel and none with another, and the reference's four captured runs do not
    locked = {n: np.array([(tail(S, s, n) < 0.2).all(0).ravel() for s in range(5)]) for n in (2, 3)}     # [SNR, lr * iter]
    for n in (2, 3):
        assert locked[n].mean() < 0.1, locked[n].sum()
        assert np.median(np.stack([tail(S, s, n).reshape(4, -1) for s in range(5)]), axis=(0, 2)).min() > 0.6
    # the reference's own runs at four grid points (one run each, lr 2.5e-3 = index 0 of the lr axis)
    for fx, tol_ser, tol_var in (("G13_cfg5_full_nu0872_snr20", 0.04, 0.03), ("G13_cfg5_full_nu1222_snr28", 0.04, 0.03),
                                 ("G13_cfg5_full_nu0271_snr26", None, 0.05), ("G13_cfg5_full_nu0_snr20", None, 0.05)):
        g = load_golden(fx)
        s, n = SNR.index(int(g["SNR"])), int(np.argmin(np.abs(np.array(NU) - float(g["nu"]))))
        ours_s, ours_v = np.median(tail(S, s, n)[:, 0], axis=-1), np.median(tail(V, s, n)[:, 0], axis=-1)     # lr 2.5e-3, median over the 5 seeds
        ref_s, ref_v = g["SER"][:, -30:].mean(1), g["Var_est"][:, -30:].mean(1)
        if tol_ser is None:                                                             # locked: Monte-Carlo error + seed-to-seed spread of the tracking error
            assert np.max(np.abs(ours_s - ref_s)) < 0.15 * ref_s.mean() + 1e-3, (fx, ours_s, ref_s)
        else:                                                                           # plateau
            assert np.max(np.abs(ours_s - ref_s)) < tol_ser, (fx, ours_s, ref_s)
        assert np.max(np.abs(ours_v - ref_v) / ref_v) < tol_var, (fx, ours_v, ref_v)


def test_eval_run_dp_batches_by_symbol_rate(tmp_path, monkeypatch):
    """A symb_rate sweep axis (Eval_run_DP.py lists [40e9 .. 100e9]) with a device generator: each rate is simulated in its own batch, so every
    result row carries the rate it is labelled with (a mixed batch used to run at the first rate)."""
    from vae_equalizer_amd import Eval_run_DP as ev
    from vae_equalizer_amd.dp_runs import DPRun, run_dp_batch
    for k, v in dict(symb_rate_vec=[40e9, 90e9], lr_optim_vec=[2.5e-3], iter=2, num_frames=4, N_frame_max=2000, generator="hip", base_seed=3,
                     savePATH=str(tmp_path) + "/").items():
        monkeypatch.setattr(ev, k, v)
    name, d = ev.main()
    S = d["SER"]                                                                        # [4, SNR, rate, nu, td, M, lr, B, fs, th, iter, frames]
    assert S.shape == (4, 1, 2, 1, 1, 1, 1, 1, 1, 1, 2, 4) and np.isfinite(S).all()
    pts = list(ev.sweep_points())
    for sr, rate in enumerate([40e9, 90e9]):
        idx = [i for i, (_, p) in enumerate(pts) if p["symb_rate"] == rate]
        runs = [DPRun(23, 0, 0.06 * np.pi, np.pi / 10, 2.5e-3, rate, 3 + 1000 * i) for i in idx]
        r = run_dp_batch(runs, "64-QAM", 2, 25, 100, 2000, 4, 10, "h0", ev.tau_cd, ev.tau_pmd, ev.phiIQ, 170, generator="hip")
        assert np.array_equal(r["SER"].numpy().transpose(1, 0, 2), S[:, 0, sr, 0, 0, 0, 0, 0, 0, 0])       # rows carry THEIR rate
    assert not np.array_equal(S[:, 0, 0], S[:, 0, 1])
    with pytest.raises(ValueError, match="symb_rate"):
        run_dp_batch([DPRun(23, 0, 0.0, 0.3, 2.5e-3, 40e9, 1), DPRun(23, 0, 0.0, 0.3, 2.5e-3, 90e9, 2)], "64-QAM", 2, 25, 100, 2000, 1, 10, "h0",
                     ev.tau_cd, ev.tau_pmd, ev.phiIQ, 170, generator="hip")
