"""GPU parity of the AWGN VAE-LE kernel (vaeq_awgn_train / vaeq_awgn_forward) against the reference's golden vectors."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
AWGN = ["G4_awgn_16qam_cfg1", "G4_awgn_64qam_pcs_free10", "G4_awgn_4qam_small"]
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _engine(g, threads=0):
    from vae_equalizer_amd.engine import AWGNEngine
    eng = AWGNEngine(1, int(g["M_est"]), g["amp_levels"], g["P"], float(g["amp_mean"]), float(g["var"]), DEV, int(g["sps"]), threads)
    eng.set_state(g["W0"], g["h0"])
    return eng


@pytest.mark.parametrize("threads", [64, 256])
@pytest.mark.parametrize("name", AWGN)
def test_awgn_teacher_forced_step(name, threads):
    """Config 1 (AWGN 16-QAM, one minibatch from seed 5) and friends: q, out, ELBO, gradients, AMSGrad update."""
    g = load_golden(name)
    B, sps = int(g["B"]), int(g["sps"])
    eng = _engine(g, threads)
    rx = torch.from_numpy(g["rx"][None, :, :B * sps]).to(DEV)
    r = eng.train(rx, B, 1, float(g["lr"]), want_q=True, want_y=True, debug_grads=True)
    torch.cuda.synchronize()
    assert relerr(_np(r["y"])[0], g["out0"]) < 2e-6
    assert np.max(np.abs(_np(r["q"])[0] - g["q0"])) < 5e-4
    assert abs(_np(r["loss"])[0, 0] - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    assert relerr(_np(r["gh"])[0], g["gh0"]) < 2e-5
    assert relerr(_np(r["gW"])[0], g["gW0"].reshape(2, -1)) < 1e-4
    t = oracle.awgn_step_grads(g["rx"][:, :B * sps], g["W0"], g["h0"], g["amp_levels"], g["P"], float(g["amp_mean"]), float(g["var"]), sps, np.float64)
    assert relerr(_np(r["gW"])[0], t["gW"].reshape(2, -1)) < max(3 * relerr(g["gW0"], t["gW"]), 2e-6)
    assert relerr(_np(r["gh"])[0], t["gh"]) < max(3 * relerr(g["gh0"], t["gh"]), 2e-6)
    # taps after the AMSGrad step; entries whose golden gradient is rounding noise (|g| < 1e-6 max|g|: the scale
    # direction of the Dirac start, see tests/test_oracle_golden.py) move by +-lr on a coin flip and are excluded
    lr = float(g["lr"])
    ok = np.abs(g["gW0"].reshape(2, -1)) > 1e-6 * np.abs(g["gW0"]).max()
    assert np.max(np.abs(_np(eng.W)[0] - g["W1"].reshape(2, -1))[ok]) < 1e-5
    assert np.max(np.abs(_np(eng.W)[0] - g["W1"].reshape(2, -1))) < 2.01 * lr
    assert np.max(np.abs(_np(eng.h)[0] - g["h1"])) < 1e-5


def test_awgn_freerun_perturbed_start():
    """3 free steps from a non-Dirac state (no noise-driven tap): losses, taps, AMSGrad max state."""
    g = load_golden("G4_awgn_4qam_small")
    B, ns = int(g["B"]), int(g["n_steps"])
    eng = _engine(g)
    r = eng.train(torch.from_numpy(g["rx"][None]).to(DEV), B, ns, float(g["lr"]))
    torch.cuda.synchronize()
    assert np.max(np.abs(_np(r["loss"])[0] - g["loss"]) / np.abs(g["loss"])) < 2e-5
    assert np.max(np.abs(_np(eng.W)[0] - g[f"W{ns}"].reshape(2, -1))) < 2e-5
    assert np.max(np.abs(_np(eng.h)[0] - g[f"h{ns}"])) < 2e-5
    assert relerr(_np(eng.xW)[0], g["vmaxW"].reshape(2, -1)) < 1e-4 and relerr(_np(eng.xh)[0], g["vmaxh"]) < 1e-4


def test_awgn_freerun_dirac_within_coinflip():
    g = load_golden("G4_awgn_64qam_pcs_free10")
    B, ns, lr = int(g["B"]), int(g["n_steps"]), float(g["lr"])
    eng = _engine(g)
    r = eng.train(torch.from_numpy(g["rx"][None]).to(DEV), B, ns, lr)
    torch.cuda.synchronize()
    assert np.max(np.abs(_np(r["loss"])[0] - g["loss"]) / np.abs(g["loss"])) < 2e-3
    assert np.max(np.abs(_np(eng.W)[0] - g[f"W{ns}"].reshape(2, -1))) < 2.5 * lr


def test_awgn_forward_validation_pass():
    """Eval-mode forward on a long block == oracle forward (normalisation by the block's own mean |y|)."""
    g = load_golden("G4_awgn_64qam_pcs_free10")
    eng = _engine(g)
    eng.set_state(g["W3"], None)
    x = g["rx"]
    q, y = eng.forward(torch.from_numpy(x[None]).to(DEV))
    qo, yo = oracle.awgn_forward(x, g["W3"], g["amp_levels"], float(g["amp_mean"]), float(g["var"]), int(g["sps"]), np.float64)
    assert relerr(_np(y)[0], yo) < 2e-6
    assert np.max(np.abs(_np(q)[0] - qo)) < 5e-4


# ------------------------------------------------------------------ wave-per-run fast path (vaeq_awgn_wave.hip)
@pytest.mark.parametrize("name", AWGN[:2])
def test_awgn_wave_teacher_forced_step(name):
    """threads=1 forces the wave kernel (B = 350 -> 3 rounds of symbol pairs per lane): same golden checks as above."""
    test_awgn_teacher_forced_step(name, 1)


def test_awgn_wave_refuses_unsupported_shape():
    from vae_equalizer_amd._native import VaeqError
    g = load_golden("G4_awgn_4qam_small")                      # B = 41 is odd: only the generic kernel takes it
    eng = _engine(g, 1)
    with pytest.raises(VaeqError):
        eng.train(torch.from_numpy(g["rx"][None]).to(DEV), int(g["B"]), 1, float(g["lr"]))


@pytest.mark.parametrize("B,M,nlev", [(26, 25, 8), (100, 25, 4), (128, 25, 8), (130, 17, 2), (254, 9, 8), (256, 25, 2), (258, 25, 8),
                                      (350, 25, 8), (384, 17, 4), (10, 9, 4), (386, 25, 8), (512, 17, 4), (514, 9, 2), (700, 25, 8),
                                      (768, 25, 2), (770, 17, 8), (1000, 25, 4), (1024, 25, 8)])
def test_awgn_wave_matches_generic(B, M, nlev):
    """Every round count (1..3), partial last rounds, all supported tap counts, and the two- / three- / four-wave variants
    (B > 384): 4 free steps, 3 runs, wave vs generic kernel."""
    from vae_equalizer_amd.engine import AWGNEngine
    rng = np.random.default_rng(B * 100 + M)
    R, steps, sps = 3, 4, 2
    amp = np.arange(-(nlev - 1), nlev, 2).astype(np.float32)
    amp /= np.sqrt(2 * np.mean(amp ** 2))
    P = rng.uniform(0.5, 1.5, (R, nlev)).astype(np.float32)
    P /= P.sum(1, keepdims=True)
    amp_mean, var = float(np.mean(np.abs(amp))), 0.02
    sym = rng.choice(amp, (R, 2, steps * B))
    rx = np.repeat(sym, sps, axis=-1).astype(np.float32)
    rx = (0.5 * rx + 0.3 * np.roll(rx, 1, -1) + 0.05 * rng.standard_normal(rx.shape)).astype(np.float32)
    W0 = (0.05 * rng.standard_normal((R, 2, M))).astype(np.float32)
    W0[:, 0, M // 2] += 1.0
    h0 = (0.05 * rng.standard_normal((R, 2, M))).astype(np.float32)
    h0[:, 0, M // 2] += 1.0
    out = []
    for threads in (256, 1):
        eng = AWGNEngine(R, M, amp, P, amp_mean, var, DEV, sps, threads)
        eng.W.copy_(torch.from_numpy(W0)); eng.h.copy_(torch.from_numpy(h0))
        r = eng.train(torch.from_numpy(rx).to(DEV), B, steps, 1e-3, want_q=True, want_y=True, debug_grads=True)
        torch.cuda.synchronize()
        out.append({k: _np(v) for k, v in r.items() if torch.is_tensor(v)} | {"W": _np(eng.W), "h": _np(eng.h), "xW": _np(eng.xW), "xh": _np(eng.xh),
                                                         "step": _np(eng.step)})
    g, w = out
    assert np.array_equal(g["step"], w["step"])
    assert np.max(np.abs(w["loss"] - g["loss"]) / np.abs(g["loss"])) < 2e-5
    assert relerr(w["y"], g["y"]) < 1e-5
    assert np.max(np.abs(w["q"] - g["q"])) < 2e-3
    assert np.max(np.abs(w["W"] - g["W"])) < 2e-5 and np.max(np.abs(w["h"] - g["h"])) < 2e-5
    assert relerr(w["gW"], g["gW"]) < 1e-3 and relerr(w["gh"], g["gh"]) < 1e-3
    assert relerr(w["xW"], g["xW"]) < 1e-3 and relerr(w["xh"], g["xh"]) < 1e-3


# ------------------------------------------------------------------ fused validation pass (vaeq_awgn_validate)
@pytest.mark.parametrize("mod,M,N", [("64-QAM", 25, 15000), ("16-QAM", 17, 4001), ("4-QAM", 9, 700), ("16-QAM", 13, 2500)])
def test_awgn_validate_matches_torch_mirror(mod, M, N):
    """forward + find_shift + SER_q in one kernel == the torch restatement of the reference's three calls on the q tensor of
    vaeq_awgn_forward; runs cover delays (TX reference shifted by -6..+7 symbols), all four quadrant rotations and an
    unconverged equaliser (SER near 1, shift found on noise)."""
    from vae_equalizer_amd import channel as ch
    from vae_equalizer_amd.engine import AWGNEngine
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import SER_q, awgn_tables, find_shift
    rng = np.random.default_rng(M * 1000 + N)
    sps, R = 2, 8
    t = awgn_tables(mod, 0.02 if mod == "64-QAM" else 0.0, 22, "h1", sps)
    rxs, ds = [], []
    for r in range(R):
        rx, d = ch.generate_data(N + 16, t["M_channel"], t["amps"], 22, t["h_channel"], sps, "cpu", t["P"], rng=np.random.default_rng(r),
                                 noise=np.random.RandomState(r))
        k = [0, 3, -6, 7, 1, -2, 0, 5][r]                                     # delay between the RX stream and the TX reference
        rx = rx[:, 2 * 8:2 * (8 + N)]
        d = d[:, 8 + k:8 + k + N]
        rot = [1, 1j, -1, -1j, 1, 1j, 1, -1][r]                               # residual quadrant rotation of the equaliser output
        c = (rx[0].numpy() + 1j * rx[1].numpy()) * rot
        rxs.append(torch.from_numpy(np.stack([c.real, c.imag]).astype(np.float32)))
        ds.append(d)
    rx, data = torch.stack(rxs).to(DEV), torch.stack(ds).to(DEV).contiguous()
    eng = AWGNEngine(R, M, t["amps"], np.tile(t["P"], (R, 1)), t["amp_mean"], t["var"], DEV, sps)
    # a rough zero-forcing start so that most runs decide mostly right: a few training epochs from the Dirac taps
    for _ in range(40):
        eng.train(rx, 350, N // 350, 5e-3)
    eng.W[6].zero_(); eng.W[6, 0, M // 2] = 1.0                               # run 6: back to the unconverged start
    ser, sh, y = eng.validate(rx, data, 21)
    q, y2 = eng.forward(rx)
    torch.cuda.synchronize()
    assert relerr(_np(y), _np(y2)) < 1e-6                                     # packed complex MACs vs the scalar FMA chain of vaeq_awgn_forward
    amp = torch.tensor(t["amps"], dtype=torch.float32, device=DEV)
    for i in range(R):
        s_ref = int(find_shift(q[i], data[i], 21, amp, t["n"]))
        assert int(sh[i]) == s_ref, (i, int(sh[i]), s_ref)
        e_ref = float(SER_q(q[i][:, 11 + s_ref:-11], data[i][:, 11:-11 - s_ref], sps, t["n"]))
        assert abs(float(ser[i]) - e_ref) <= 2.0 / N, (i, float(ser[i]), e_ref)
    if N >= 2500:                                                             # enough training above: most delayed runs locked and were realigned
        assert int((ser[:6] < 0.05).sum()) >= 3 and {int(v) for v in sh[:6]} != {0}, (ser, sh)


def test_awgn_validate_rejects_bad_shapes():
    from vae_equalizer_amd._native import VaeqError
    from vae_equalizer_amd.engine import AWGNEngine
    eng = AWGNEngine(2, 25, np.array([-1.0, 1.0], np.float32), np.full((2, 2), 0.5, np.float32), 1.0, 0.01, DEV, 2)
    x = torch.zeros(2, 2, 2 * 32, device=DEV)
    with pytest.raises(VaeqError):
        eng.validate(x, torch.zeros(2, 2, 32, dtype=torch.float16, device=DEV))          # N < 64
    with pytest.raises(ValueError):
        eng.validate(torch.zeros(2, 2, 2 * 100, device=DEV), torch.zeros(2, 2, 99, dtype=torch.float16, device=DEV))


# ------------------------------------------------------------------ stand-alone operator mirrors of the AWGN modules
def test_awgn_operator_mirrors_against_golden():
    """func_VAELE_MQAM_shaping.twoFIR / loss_function and func_VAENN_MQAM.loss_function (values) on the captured minibatches."""
    from vae_equalizer_amd import func_VAELE_MQAM_shaping as aw
    from vae_equalizer_amd import func_VAENN_MQAM as nn_
    for name in AWGN:
        g = load_golden(name)
        B, sps = int(g["B"]), int(g["sps"])
        x = torch.from_numpy(g["rx"][:, :B * sps]).to(DEV)
        amp = torch.from_numpy(g["amp_levels"]).to(DEV)
        net = aw.twoFIR(int(g["M_est"]), sps).to(DEV)
        with torch.no_grad():
            net.conv_w.weight.copy_(torch.from_numpy(g["W0"]).reshape(1, 2, -1))
        q, out = net(x, amp, float(g["amp_mean"]), float(g["var"]))
        assert relerr(_np(out), g["out0"]) < 2e-6 and np.max(np.abs(_np(q) - g["q0"])) < 5e-4
        loss = aw.loss_function(torch.from_numpy(g["q0"]).to(DEV), x, torch.from_numpy(g["h0"]).to(DEV), DEV, amp, torch.from_numpy(g["P"]).to(DEV))
        assert abs(float(loss) - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    for name in ("G8_vaenn_64qam", "G8_vaenn_16qam_small", "G8_vaenn_4qam_k5"):
        g = load_golden(name)
        B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
        x = torch.from_numpy(g["rx"][:, :B * sps]).to(DEV)
        h0 = torch.from_numpy(g["theta0"][-2 * M:].reshape(2, M)).to(DEV)
        loss = nn_.loss_function(torch.from_numpy(g["q0"]).to(DEV), x, h0, DEV, torch.from_numpy(g["amp_levels"]).to(DEV))
        assert abs(float(loss) - g["loss"][0]) / abs(g["loss"][0]) < 1e-5


# ------------------------------------------------------------------ validation on a clean frame, noise added while it is read (vaeq_awgn_validate_gen)
@pytest.mark.parametrize("mod,M,N,fixed", [("64-QAM", 25, 15000, False), ("16-QAM", 17, 4001, False), ("4-QAM", 9, 2050, True), ("64-QAM", 25, 1022, False)])
def test_awgn_validate_on_clean_frame_equals_generate_then_validate(mod, M, N, fixed):
    """vaeq_gen_awgn_clean + vaeq_awgn_validate_gen == vaeq_gen_awgn + vaeq_awgn_validate for the same (seed, frame), BIT FOR BIT: the same
    symbols and TX reference, the same noise level, the same noisy samples (y is the equaliser's output on them), hence the same shift and SER.
    Frame lengths that are no multiple of the 1024-symbol tile or of four samples, power-derived and fixed noise levels, per-run SNRs."""
    from vae_equalizer_amd import channel as ch
    from vae_equalizer_amd.engine import AWGNEngine
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    sps, R = 2, 6
    t = awgn_tables(mod, 0.02 if mod == "64-QAM" else 0.0, 22, "h1", sps)
    snr = np.array([14, 18, 22, 26, 30, 22], np.float32)
    sf = (0.05 + 0.01 * np.arange(R)).astype(np.float32) if fixed else None
    eng = AWGNEngine(R, M, t["amps"], np.tile(t["P"], (R, 1)), t["amp_mean"], t["var"], DEV, sps)
    rx0, _ = ch.generate_awgn_batch_hip(R, 4200, t["amps"], t["P"], snr, t["h_channel"], sps, DEV, 91, 0)
    for _ in range(25):                                                       # some training so that the equaliser is no Dirac
        eng.train(rx0, 350, 12, 5e-3)
    for frame in (1, 7):
        rx, data, sigma = ch.generate_awgn_batch_hip(R, N, t["amps"], t["P"], snr, t["h_channel"], sps, DEV, 91, frame, return_sigma=True, sigma_fixed=sf)
        ser_a, sh_a, y_a = eng.validate(rx, data, 21)
        clean = ch.generate_awgn_clean_batch_hip(R, N, t["amps"], t["P"], snr, t["h_channel"], sps, DEV, 91, frame, sigma_fixed=sf)
        ser_b, sh_b, y_b, sigma_b = eng.validate_clean(clean, 21, return_sigma=True)
        torch.cuda.synchronize()
        assert torch.equal(clean.data, data)
        assert torch.equal(sigma_b, sigma)
        assert torch.equal(y_b, y_a)
        assert torch.equal(sh_b, sh_a) and torch.equal(ser_b, ser_a)
        # and the clean frame is what the noisy one was made from: their difference is noise of the requested level
        c = clean.sig[:, :sps * N].cpu().numpy().astype(np.float64)
        nz = rx.cpu().numpy().astype(np.float64) - np.stack([c[..., 0], c[..., 1]], 1)
        s = sigma.cpu().numpy()
        assert np.all(np.abs(nz.std(axis=2) / s[:, None] - 1) < 0.06), (nz.std(axis=2), s)
    assert float(ser_a.min()) < 0.2                                           # the comparison is not between two garbage outputs


def test_awgn_sweep_uses_the_clean_validation_frame_and_matches_the_two_step_form(monkeypatch):
    """run_awgn_batch with the device generator validates on a clean frame (noise added on load); forcing the two-step form gives the same
    SER_valid bit for bit (same seed: same Philox draws in both forms)."""
    from vae_equalizer_amd import channel as ch
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import run_awgn_batch
    runs = [dict(SNR=s, nu=0.0270955, lr_optim=lr) for s in (20, 24) for lr in (3e-3, 5e-3)]
    kw = dict(mod="64-QAM", sps=2, M_est=25, batch_len=350, N_valid=3000, N_train=1200, num_epochs=8, epe=2, channel="h1", device=DEV,
              generator="hip", seed=1234)
    calls = {"n": 0}
    orig = ch.generate_awgn_clean_batch_hip
    monkeypatch.setattr(ch, "generate_awgn_clean_batch_hip", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), orig(*a, **k))[1])
    a = run_awgn_batch(runs, **kw)
    assert calls["n"] == 4
    monkeypatch.setattr(ch, "awgn_clean_supported", lambda sps, M: False)
    b = run_awgn_batch(runs, **kw)
    assert calls["n"] == 4 and torch.equal(a, b)



def test_awgn_validate_clean_refuses_what_it_cannot_do():
    """The noise-on-load form exists for sps = 2 and the baked tap counts: anything else is refused loudly (VAEQ_ERR_SHAPE), never silently
    computed another way; run_awgn_batch asks channel.awgn_clean_supported first and takes the two-step form otherwise."""
    from vae_equalizer_amd import _native as nat, channel as ch
    from vae_equalizer_amd.engine import AWGNEngine
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import awgn_tables
    t = awgn_tables("16-QAM", 0.0, 20, "h1", 2)
    R, N = 3, 2000
    clean = ch.generate_awgn_clean_batch_hip(R, N, t["amps"], t["P"], 20.0, t["h_channel"], 2, DEV, 5, 0)
    assert ch.awgn_clean_supported(2, 25) and not ch.awgn_clean_supported(2, 13) and not ch.awgn_clean_supported(3, 25)
    eng13 = AWGNEngine(R, 13, t["amps"], np.tile(t["P"], (R, 1)), t["amp_mean"], t["var"], DEV, 2)
    with pytest.raises(nat.VaeqError, match="unsupported sizes"):
        eng13.validate_clean(clean, 21)
    eng4 = AWGNEngine(R + 1, 25, t["amps"], np.tile(t["P"], (R + 1, 1)), t["amp_mean"], t["var"], DEV, 2)
    with pytest.raises(ValueError):
        eng4.validate_clean(clean, 21)
    with pytest.raises(ValueError):
        ch.generate_awgn_clean_batch_hip(R, N, t["amps"], t["P"], 20.0, awgn_tables("16-QAM", 0.0, 20, "h1", 3)["h_channel"], 3, DEV, 5, 0)
