#!/usr/bin/env python3
"""Capture golden input/output vectors from the reference (kit-cel/vae-equalizer).

Runs ONLY in the development container, where the read-only reference tree is
mounted at /root/reference.  It imports the reference's Python modules as they
are, feeds them seeded inputs and writes small ``.npz`` fixtures (inputs AND
expected outputs) to ``tests/golden/``.  Nothing from the reference is copied:
fixtures are data.

Two version-drift shims are applied from the outside (the reference pins
numpy==1.18.4; this image has numpy 2.2), see SURVEY.md section 8c:
  * ``numpy.core.numeric.Inf`` no longer exists        -> alias of ``numpy.inf``
  * ``np.asarray([[arr, 0], [0, arr]])`` is now ragged  -> ``simulate_dispersion``
    is replaced by an explicit per-frequency 2x2 product with identical math.
The reference seeds nothing; determinism comes from patching
``np.random.default_rng`` / ``np.random.seed`` / ``torch.manual_seed`` here.

Fixture families: G0 init tables, G1-G3 DP step / free runs (VAE-LE, VAEflex), G4 AWGN VAE-LE, G5 DP epilogue, G6 generator,
G7 processing()-level runs (configs 1-3), G8 AWGN VAE-NN (Net), G9 converging VAEflex run (config 4), G10 converging PCS VAE-LE
run (config 5 shape), G11 VAE-NN with BatchNorm (Net_BN), G12 CMA / CPE, G1b heavy-shaping DP steps, G13 config-5 on-grid runs,
G14 the CMA modules' epilogue at 16- / 64-QAM.

Usage:  python tools/capture_golden.py [--only G1,G2] [--full-run]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

NU_572 = 0.0270955  # PCS-64-QAM, H = 5.72 bit (Eval_run_DP.py:24)


# --------------------------------------------------------------------------
# reference import with the two numpy-2 shims
# --------------------------------------------------------------------------
def _import_reference():
    import numpy.core.numeric as _ncn  # noqa: deprecated alias, needed for the shim

    if not hasattr(_ncn, "Inf"):
        _ncn.Inf = np.inf
    sys.path.insert(0, os.path.join(REF, "optical_DP_channel"))
    sys.path.insert(0, os.path.join(REF, "AWGN_channel"))
    import shared_funcs as sfun  # reference module
    import func_VAELE_MQAM_shaping as awgn  # reference module

    def simulate_dispersion(rx, symb_rate, sps, tau_cd, tau_pmd, phiIQ, theta):
        # same math as shared_funcs.py:38-54 with the 2x2 matrix product written
        # out per frequency bin (numpy>=1.24 rejects the ragged asarray at :49)
        rx_fft = np.fft.fft(rx, axis=1)
        freq = np.fft.fftfreq(rx.shape[1], 1 / symb_rate / sps)
        exp_cd, exp_pmd = np.exp(1j * 2 * (np.pi * freq) ** 2 * tau_cd), np.exp(1j * np.pi * tau_pmd * freq)
        c, s = np.cos(theta), np.sin(theta)
        e = np.exp(-1j * phiIQ)
        R = np.asarray([[c * e[0], s * e[0]], [-s * e[1], c * e[1]]])
        RT = np.asarray([[c * e[0], -s * e[0]], [s * e[1], c * e[1]]])
        d0, d1 = exp_pmd, 1 / exp_pmd
        H00 = RT[0, 0] * d0 * R[0, 0] + RT[0, 1] * d1 * R[1, 0]
        H01 = RT[0, 0] * d0 * R[0, 1] + RT[0, 1] * d1 * R[1, 1]
        H10 = RT[1, 0] * d0 * R[0, 0] + RT[1, 1] * d1 * R[1, 0]
        H11 = RT[1, 0] * d0 * R[0, 1] + RT[1, 1] * d1 * R[1, 1]
        out = np.zeros((2, rx.shape[1]), dtype=np.complex128)
        out[0] = (H00 * rx_fft[0] + H01 * rx_fft[1]) * exp_cd
        out[1] = (H10 * rx_fft[0] + H11 * rx_fft[1]) * exp_cd
        return np.complex64(np.fft.ifft(out, axis=1))

    sfun.simulate_dispersion = simulate_dispersion
    return sfun, awgn


class SeededRng:
    """Patch for ``np.random.default_rng``: call k returns Generator(seed + k)."""

    def __init__(self, seed):
        self.seed, self.k = seed, 0
        self._orig = np.random.default_rng

    def __call__(self, *a, **kw):
        g = self._orig(self.seed + self.k)
        self.k += 1
        return g

    def __enter__(self):
        np.random.default_rng = self
        np.random.seed(self.seed)
        return self

    def __exit__(self, *exc):
        np.random.default_rng = self._orig


DP_DEFAULTS = dict(symb_rate=90e9, tau_cd=-26e-24, tau_pmd=0.1e-12 * np.sqrt(1000),
                   phiIQ=np.array([0.0314, 0.0314], dtype=np.complex64), theta=np.pi / 10)


def t2n(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------
# G0: init() tables
# --------------------------------------------------------------------------
def capture_G0(sfun, awgn):
    out = {}
    k = 0
    for mod in ("4-QAM", "16-QAM", "64-QAM"):
        for nu in (0.0, NU_572, 0.0872449, 0.1222578):
            for SNR in (20, 23):
                for channel, M_est in (("h0", 25), ("h1", 13)):
                    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init(
                        channel, mod, "cpu", nu, 2, M_est, SNR)
                    pre = f"c{k}_"
                    out[pre + "args"] = np.array([mod, str(nu), str(SNR), channel, str(M_est)])
                    out[pre + "h_est"] = t2n(h_est)
                    out[pre + "h_channel"] = h_ch
                    out[pre + "P"] = P
                    out[pre + "amp_levels"] = t2n(amp_levels)
                    out[pre + "amps"] = amps
                    out[pre + "nu_sc"] = np.float64(nu_sc)
                    out[pre + "var"] = t2n(var)
                    out[pre + "pow_mean"] = np.float64(pow_mean)
                    k += 1
    out["n_cases"] = np.int64(k)
    save("G0_init", **out)


# --------------------------------------------------------------------------
# G1: DP teacher-forced single steps (forward, loss, autograd grads, 3 Adam steps)
# --------------------------------------------------------------------------
def _dp_case(sfun, mod, nu, SNR, M_est, B, seed, lr=2.5e-3, perturb=0.05, channel="h0", n_steps=3):
    sps = 2
    torch.manual_seed(seed)
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init(channel, mod, "cpu", nu, sps, M_est, SNR)
    P_t = torch.tensor(P, dtype=torch.float32)
    with SeededRng(seed):
        rx, data, sigma_n = sfun.generate_data_shaping(B * n_steps, amps, SNR, h_ch, P, pol, DP_DEFAULTS["symb_rate"], sps,
                                                       DP_DEFAULTS["tau_cd"], DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"],
                                                       DP_DEFAULTS["theta"], "cpu")
    net = sfun.twoXtwoFIR(M_est, sps)
    with torch.no_grad():
        net.conv_w.weight.add_(perturb * torch.randn_like(net.conv_w.weight))
        h_est.add_(perturb * torch.randn_like(h_est))
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    opt.add_param_group({"params": h_est})
    res = dict(rx=t2n(rx), W0=t2n(net.conv_w.weight), h0=t2n(h_est), amp_levels=t2n(amp_levels), P=P_t.numpy().copy(),
               var=t2n(var), nu_sc=np.float64(nu_sc), lr=np.float64(lr), B=np.int64(B), M_est=np.int64(M_est),
               sps=np.int64(sps), n_steps=np.int64(n_steps), mod=np.array(mod), nu=np.float64(nu), SNR=np.float64(SNR))
    for s in range(n_steps):
        mb = rx[:, :, s * B * sps:(s + 1) * B * sps].clone()
        opt.zero_grad()
        q, out = net(mb, amp_levels, var, nu_sc)
        loss, var_est = sfun.loss_function_shaping(q.squeeze(), mb.squeeze(), h_est, amp_levels, P_t)
        loss.backward()
        res[f"q{s}"], res[f"out{s}"] = t2n(q), t2n(out)
        res[f"loss{s}"], res[f"var_est{s}"] = t2n(loss), t2n(var_est)
        res[f"gW{s}"], res[f"gh{s}"] = t2n(net.conv_w.weight.grad), t2n(h_est.grad)
        opt.step()
        res[f"W{s + 1}"], res[f"h{s + 1}"] = t2n(net.conv_w.weight), t2n(h_est)
    st = opt.state[net.conv_w.weight]
    res["mW"], res["vW"] = t2n(st["exp_avg"]), t2n(st["exp_avg_sq"])
    st = opt.state[h_est]
    res["mh"], res["vh"] = t2n(st["exp_avg"]), t2n(st["exp_avg_sq"])
    return res


def capture_G1(sfun, awgn):
    save("G1_dp_step_64qam_pcs", **_dp_case(sfun, "64-QAM", NU_572, 23, 25, 100, seed=11))
    save("G1_dp_step_64qam", **_dp_case(sfun, "64-QAM", 0.0, 23, 25, 100, seed=12))
    save("G1_dp_step_16qam", **_dp_case(sfun, "16-QAM", 0.0872449, 20, 13, 50, seed=13, channel="h1"))
    save("G1_dp_step_4qam", **_dp_case(sfun, "4-QAM", 0.0, 14, 9, 37, seed=14, lr=1e-3))


# --------------------------------------------------------------------------
# G2 / G3: DP free runs through the reference's own loops (VAE-LE and VAEflex)
# --------------------------------------------------------------------------
def _dp_free_run(sfun, flex, n_steps, seed, mod="64-QAM", nu=0.0, SNR=23, M_est=25, B=100, flex_step=10, lr=2.5e-3):
    sps = 2
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", mod, "cpu", nu, sps, M_est, SNR)
    n = amp_levels.shape[0]
    P_t = torch.tensor(P, dtype=torch.float32)
    N_sym = n_steps * B if not flex else (n_steps - 1) * flex_step + B
    with SeededRng(seed):
        rx, data, sigma_n = sfun.generate_data_shaping(N_sym, amps, SNR, h_ch, P, pol, DP_DEFAULTS["symb_rate"], sps,
                                                       DP_DEFAULTS["tau_cd"], DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"],
                                                       DP_DEFAULTS["theta"], "cpu")
    net = sfun.twoXtwoFIR(M_est, sps)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    opt.add_param_group({"params": h_est})
    keep = B if not flex else flex_step
    out_train = torch.empty(2, 2 * n, n_steps * keep)
    out_const = torch.empty(2, 2, n_steps * keep)
    var_est = torch.empty(2, n_steps)
    losses = np.zeros(n_steps, dtype=np.float32)
    snaps = {}
    for s in range(n_steps):
        m = s * (B if not flex else flex_step)
        mb = rx[:, :, m * sps:(m + B) * sps].clone()
        opt.zero_grad()
        q, out = net(mb, amp_levels, var, nu_sc)
        k0 = 0 if not flex else (B - flex_step) // 2
        out_train[:, :, s * keep:(s + 1) * keep] = q[:, :, k0:k0 + keep].detach()
        out_const[:, :, s * keep:(s + 1) * keep] = out[:, :, k0:k0 + keep].detach()
        loss, var_est[:, s] = sfun.loss_function_shaping(q.squeeze(), mb.squeeze(), h_est, amp_levels, P_t)
        loss.backward()
        opt.step()
        losses[s] = loss.item()
        if (s + 1) in (1, 5, 10, 20, n_steps):
            snaps[f"W_after{s + 1}"], snaps[f"h_after{s + 1}"] = t2n(net.conv_w.weight), t2n(h_est)
    return dict(rx=t2n(rx), data=t2n(data), amp_levels=t2n(amp_levels), P=P_t.numpy().copy(), var=t2n(var),
                nu_sc=np.float64(nu_sc), lr=np.float64(lr), B=np.int64(B), M_est=np.int64(M_est), sps=np.int64(sps),
                flex_step=np.int64(flex_step if flex else B), n_steps=np.int64(n_steps), loss=losses,
                var_est=t2n(var_est), out_train=t2n(out_train), out_const=t2n(out_const), **snaps)


def capture_G2(sfun, awgn):
    save("G2_dp_freerun", **_dp_free_run(sfun, False, 30, seed=21))


def capture_G3(sfun, awgn):
    save("G3_dp_flex_freerun", **_dp_free_run(sfun, True, 30, seed=31, nu=NU_572))


# --------------------------------------------------------------------------
# G4: AWGN VAE-LE (twoFIR + loss_function + Adam amsgrad)
# --------------------------------------------------------------------------
def _awgn_tables(mod, nu, SNR, channel="h1", sps=2):
    # constants exactly as processing() builds them, func_VAELE_MQAM_shaping.py:239-272;
    # obtained by running the reference's own expressions on the reference's tables.
    consts = {"4-QAM": 2, "16-QAM": 4, "64-QAM": 8}
    nlev = consts[mod]
    lev = np.arange(-(nlev - 1), nlev, 2).astype(np.float64)
    const = (lev[:, None] + 1j * lev[None, :]).reshape(-1)
    const = const / np.sqrt(np.mean(np.abs(const) ** 2))
    amps = const.real[::nlev]
    sc = np.min(np.abs(amps))
    P = np.exp(-nu * np.abs(amps / sc) ** 2)
    P = P / np.sum(P)
    sm = np.tile(P, (nlev, 1))
    sm = (sm * sm.T).reshape(-1) * const
    amp_mean = np.sum(np.abs(sm.real) + np.abs(sm.imag)) / 2
    var = 10 ** (-SNR / 10)
    h_orig = {"h1": np.array([0.0545 + 1j * 0.05, 0.2823 - 1j * 0.11971, -0.7676 + 1j * 0.2788, -0.0641 - 1j * 0.0576,
                              0.0466 - 1j * 0.02275]),
              "h2": np.array([0.0545 + 1j * 0.0165, -1.3449 - 1j * 0.4523, 1.0067 + 1j * 1.1524, 0.3476 + 1j * 0.3153])}[channel]
    h_orig = h_orig.astype(np.complex64)
    h_ch = np.zeros(sps * (len(h_orig) - 1) + 1, dtype=np.complex64)
    h_ch[0::sps] = h_orig
    h_ch /= np.linalg.norm(h_ch)
    return amps, P, amp_mean, var, h_ch, len(h_orig)


def _awgn_case(awgn, mod, nu, SNR, M_est, B, seed, n_steps, lr=5e-3, perturb=0.0):
    sps = 2
    torch.manual_seed(seed)
    amps, P, amp_mean, var, h_ch, Mch = _awgn_tables(mod, nu, SNR)
    amp_levels = torch.tensor(amps, dtype=torch.float32)
    P_t = torch.tensor(P, dtype=torch.float32)
    with SeededRng(seed):
        rx, data = awgn.generate_data(B * n_steps, Mch, amps, SNR, h_ch, sps, "cpu", P)
    net = awgn.twoFIR(M_est, sps)
    h_est = np.zeros([2, M_est])
    h_est[0, M_est // 2] = 1
    h_est = torch.tensor(h_est, requires_grad=True, dtype=torch.float32)
    if perturb:
        with torch.no_grad():
            net.conv_w.weight.add_(perturb * torch.randn_like(net.conv_w.weight))
            h_est.add_(perturb * torch.randn_like(h_est))
    opt = torch.optim.Adam(net.parameters(), lr=lr, amsgrad=True)
    opt.add_param_group({"params": h_est})
    res = dict(rx=t2n(rx), data=t2n(data), W0=t2n(net.conv_w.weight), h0=t2n(h_est), amp_levels=t2n(amp_levels),
               P=P_t.numpy().copy(), amp_mean=np.float64(amp_mean), var=np.float64(var), lr=np.float64(lr),
               B=np.int64(B), M_est=np.int64(M_est), sps=np.int64(sps), n_steps=np.int64(n_steps), mod=np.array(mod),
               nu=np.float64(nu), SNR=np.float64(SNR))
    losses = np.zeros(n_steps, dtype=np.float32)
    for s in range(n_steps):
        mb = rx[:, s * B * sps:(s + 1) * B * sps]
        opt.zero_grad()
        q, out = net(mb, amp_levels, amp_mean, var)
        loss = awgn.loss_function(q, mb, h_est, "cpu", amp_levels, P_t)
        loss.backward()
        losses[s] = loss.item()
        if s < 3:
            res[f"q{s}"], res[f"out{s}"] = t2n(q), t2n(out)
            res[f"gW{s}"], res[f"gh{s}"] = t2n(net.conv_w.weight.grad), t2n(h_est.grad)
        opt.step()
        if s < 3 or s + 1 == n_steps:
            res[f"W{s + 1}"], res[f"h{s + 1}"] = t2n(net.conv_w.weight), t2n(h_est)
    res["loss"] = losses
    st = opt.state[net.conv_w.weight]
    res["mW"], res["vW"], res["vmaxW"] = t2n(st["exp_avg"]), t2n(st["exp_avg_sq"]), t2n(st["max_exp_avg_sq"])
    st = opt.state[h_est]
    res["mh"], res["vh"], res["vmaxh"] = t2n(st["exp_avg"]), t2n(st["exp_avg_sq"]), t2n(st["max_exp_avg_sq"])
    return res


def capture_G4(sfun, awgn):
    # config 1: AWGN 16-QAM, nu=0, one minibatch from seed 5 (+2 more teacher-free steps), SNR 24, B=350
    save("G4_awgn_16qam_cfg1", **_awgn_case(awgn, "16-QAM", 0.0, 24, 25, 350, seed=5, n_steps=3))
    # config 2 shape: 64-QAM + PCS, 10-step free run from Dirac
    save("G4_awgn_64qam_pcs_free10", **_awgn_case(awgn, "64-QAM", NU_572, 24, 25, 350, seed=6, n_steps=10))
    # perturbed state, small odd sizes
    save("G4_awgn_4qam_small", **_awgn_case(awgn, "4-QAM", 0.0, 12, 9, 41, seed=7, n_steps=3, perturb=0.05, lr=1e-3))


# --------------------------------------------------------------------------
# G5: per-frame epilogue (shift search + both SER estimators) on a converged frame;
#     also a short SER trajectory of the reference's DP VAE-LE loop at a reduced frame size.
# --------------------------------------------------------------------------
def capture_G5(sfun, awgn, num_frames=140, N_frame_max=1000, B=100, seed=51, mod="64-QAM", nu=0.0, SNR=23, M_est=25,
               lr=2.5e-3, theta_diff=0.006 * np.pi):
    # Drives the reference's own primitives in the order of func_VAELE_DP_MQAM_shaping.py:43-89
    # (processing() itself prints but does not expose the per-frame tensors we need to pin).
    sps, N_cut = 2, 10
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", mod, "cpu", nu, sps, M_est, SNR)
    n = amp_levels.shape[0]
    P_t = torch.tensor(P, dtype=torch.float32)
    net = sfun.twoXtwoFIR(M_est, sps)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    opt.add_param_group({"params": h_est})
    m_max = N_frame_max // B
    N_frame = m_max * B
    theta = DP_DEFAULTS["theta"]
    SER_valid = torch.empty(4, num_frames)
    Var_est = torch.empty(2, num_frames)
    shifts = np.zeros((num_frames, 2, 2), dtype=np.int64)
    rs = np.zeros((num_frames, 2), dtype=np.int64)
    keep = {}
    t0 = time.time()
    with SeededRng(seed):
        for frame in range(num_frames):
            rx, data, _ = sfun.generate_data_shaping(N_frame, amps, SNR, h_ch, P, pol, DP_DEFAULTS["symb_rate"], sps,
                                                     DP_DEFAULTS["tau_cd"], DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"], theta, "cpu")
            theta += theta_diff
            out_train = torch.empty(pol, 2 * n, N_frame)
            out_const = torch.empty(pol, 2, N_frame)
            var_est = torch.empty(pol, m_max)
            for m in range(m_max):
                mb = rx[:, :, m * B * sps:(m + 1) * B * sps].clone()
                opt.zero_grad()
                q, out = net(mb, amp_levels, var, nu_sc)
                out_train[:, :, m * B:(m + 1) * B] = q.detach().clone()
                out_const[:, :, m * B:(m + 1) * B] = out.detach().clone()
                loss, var_est[:, m] = sfun.loss_function_shaping(q.squeeze(), mb.squeeze(), h_est, amp_levels, P_t)
                loss.backward()
                opt.step()
            Var_est[:, frame] = torch.mean(var_est, dim=1)
            last = frame == num_frames - 1
            if last:
                keep.update(out_train=t2n(out_train), out_const=t2n(out_const), data=t2n(data))
            shift, r = sfun.find_shift(out_train, data, 21, amp_levels, pol)
            shifts[frame, 0], rs[frame, 0] = shift.numpy(), r
            ot = out_train.roll(r, 0)
            ot[0, :, :], ot[1, :, :] = ot[0, :, :].roll(int(-shift[0]), -1), ot[1, :, :].roll(int(-shift[1]), -1)
            ot = ot.reshape(pol, 2 * n, m_max, B)[:, :, :, :B - shift[0] - N_cut].reshape(pol, 2 * n, -1)
            dt = data.reshape(pol, 2, m_max, B)[:, :, :, :B - shift[0] - N_cut].reshape(pol, 2, -1)
            ms = torch.max(torch.abs(shift))
            SER_valid[2:, frame] = sfun.SER_IQflip(ot[:, :, 11:-11 - ms], dt[:, :, 11:-11 - ms])
            shift, r = sfun.find_shift_symb_full(out_const, data, 21)
            shifts[frame, 1], rs[frame, 1] = shift.numpy(), r
            oc = out_const.roll(r, 0)
            oc[0, :, :], oc[1, :, :] = oc[0, :, :].roll(int(-shift[0]), -1), oc[1, :, :].roll(int(-shift[1]), -1)
            oc = oc.reshape(pol, 2, m_max, B)[:, :, :, :B - shift[0] - N_cut].reshape(pol, 2, -1)
            dt = data.reshape(pol, 2, m_max, B)[:, :, :, :B - shift[0] - N_cut].reshape(pol, 2, -1)
            ms = torch.max(torch.abs(shift))
            SER_valid[:2, frame] = sfun.SER_constell_shaping(oc[:, :, 11:-11 - ms].detach().clone(), dt[:, :, 11:-11 - ms],
                                                            amp_levels, nu_sc, var)
            print(f"   G5 frame {frame}: loss {loss.item():.2f} SER {SER_valid[:, frame].tolist()} shift {shifts[frame].tolist()} "
                  f"r {rs[frame].tolist()}  [{time.time() - t0:.0f}s]", flush=True)
    save("G5_dp_epilogue", amp_levels=t2n(amp_levels), var=t2n(var), nu_sc=np.float64(nu_sc), P=P_t.numpy().copy(),
         B=np.int64(B), N_frame_max=np.int64(N_frame_max), num_frames=np.int64(num_frames), seed=np.int64(seed),
         lr=np.float64(lr), theta_diff=np.float64(theta_diff), SNR=np.float64(SNR), nu=np.float64(nu),
         M_est=np.int64(M_est), mod=np.array(mod), pow_mean=np.float64(pow_mean),
         SER_valid=t2n(SER_valid), Var_est=t2n(Var_est), shifts=shifts, rs=rs, **keep)


# --------------------------------------------------------------------------
# G6: channel generator (rrcfir, generate_data_shaping) with the seeded patch
# --------------------------------------------------------------------------
def capture_G6(sfun, awgn):
    res = dict(rrc_8_2_01=sfun.rrcfir(8, 2, 0.1), rc_8_2_01=sfun.rcfir(8, 2, 0.1))
    for tag, (mod, nu, SNR, channel) in dict(a=("64-QAM", NU_572, 23, "h0"), b=("16-QAM", 0.0, 18, "h1")).items():
        h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init(channel, mod, "cpu", nu, 2, 25, SNR)
        with SeededRng(61):
            rx, data, sigma_n = sfun.generate_data_shaping(300, amps, SNR, h_ch, P, pol, DP_DEFAULTS["symb_rate"], 2,
                                                           DP_DEFAULTS["tau_cd"], DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"],
                                                           0.3, "cpu")
            rx2, data2, _ = sfun.generate_data_shaping(300, amps, SNR, h_ch, P, pol, DP_DEFAULTS["symb_rate"], 2,
                                                       DP_DEFAULTS["tau_cd"], DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"],
                                                       0.5, "cpu")
        res.update({f"{tag}_rx": t2n(rx), f"{tag}_data": t2n(data), f"{tag}_sigma_n": np.float64(sigma_n),
                    f"{tag}_rx2": t2n(rx2), f"{tag}_data2": t2n(data2),
                    f"{tag}_args": np.array([mod, str(nu), str(SNR), channel])})
    # AWGN generator
    amps, P, amp_mean, var, h_ch, Mch = _awgn_tables("16-QAM", 0.0, 24)
    with SeededRng(62):
        rx, data = awgn.generate_data(200, Mch, amps, 24, h_ch, 2, "cpu", P)
    res.update(awgn_rx=t2n(rx), awgn_data=t2n(data))
    save("G6_generator", **res)


# --------------------------------------------------------------------------
# G7: statistics of full reference runs through processing() itself
# --------------------------------------------------------------------------
def capture_G7(sfun, awgn, full=False):
    import contextlib
    import io
    import func_VAELE_DP_MQAM_shaping as ref_vaele
    import func_VAEflex_DP_MQAM_shaping as ref_flex

    res = {}
    # (a) DP VAE-LE through processing() itself: same reduced config and seed as G5 (140 frames x 1000
    #     symbols, converges around frame 110), so G5's hand-driven loop and processing() pin each other;
    #     VAEflex: 6 frames x 400 symbols (30 window steps per frame)
    for tag, mod_, kw in (("vaele", ref_vaele, dict(N_frame_max=1000, num_frames=140, flex_step=10, seed=51, td=0.006 * np.pi)),
                          ("flex", ref_flex, dict(N_frame_max=400, num_frames=6, flex_step=10, seed=71, td=0.06 * np.pi))):
        t0 = time.time()
        with SeededRng(kw["seed"]), contextlib.redirect_stdout(io.StringIO()):
            SER, Var_est, var = mod_.processing("64-QAM", 2, 23, 0.0, 25, kw["td"], np.pi / 10, 2.5e-3, 100,
                                                kw["N_frame_max"], kw["num_frames"], kw["flex_step"], "h0", 90e9, -26e-24,
                                                0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], dtype=np.complex64), 170)
        print(f"   G7 {tag}: {time.time() - t0:.0f}s  SER last {SER[:, -1].tolist()}")
        res.update({f"{tag}_SER": t2n(SER), f"{tag}_Var_est": t2n(Var_est), f"{tag}_var": t2n(var),
                    f"{tag}_N_frame_max": np.int64(kw["N_frame_max"]), f"{tag}_num_frames": np.int64(kw["num_frames"]),
                    f"{tag}_seed": np.int64(kw["seed"]), f"{tag}_theta_diff": np.float64(kw["td"])})
    # (b) AWGN processing(): config-1 shape, 20 epochs
    t0 = time.time()
    with SeededRng(72), contextlib.redirect_stdout(io.StringIO()):
        SERa = awgn.processing("16-QAM", 2, 24, 0.0, 25, 5e-3, 350, 15000, 1200, 20, 2, "h1")
    print(f"   G7 awgn: {time.time() - t0:.0f}s  SER {SERa.tolist()}")
    res["awgn_SER"] = t2n(SERa)
    if full:
        t0 = time.time()
        with SeededRng(1234), contextlib.redirect_stdout(io.StringIO()):
            SER, Var_est, var = ref_vaele.processing("64-QAM", 2, 23, 0.0, 25, 0.06 * np.pi, np.pi / 10, 2.5e-3, 100, 10000,
                                                     170, 10, "h0", 90e9, -26e-24, 0.1e-12 * np.sqrt(1000),
                                                     np.array([0.0314, 0.0314], dtype=np.complex64), 170)
        print(f"   G7 full: {time.time() - t0:.0f}s  SER last {SER[:, -1].tolist()}")
        res.update(full_SER=t2n(SER), full_Var_est=t2n(Var_est), full_var=t2n(var), full_seconds=np.float64(time.time() - t0))
    save("G7_full_runs" if full else "G7_runs", **res)


# --------------------------------------------------------------------------
# G8: AWGN VAE-NN (row f3): Net + loss_function + Adam(amsgrad) of AWGN_channel/func_VAENN_MQAM.py
# --------------------------------------------------------------------------
def _nn_case(mod, SNR, M_est, k1, k2, B, seed, n_steps, lr=4e-3, channel="h1"):
    import func_VAENN_MQAM as nnref  # reference module

    sps = 2
    torch.manual_seed(seed)
    ir = {"h1": [0.0545 + 0.05j, 0.2823 - 0.11971j, -0.7676 + 0.2788j, -0.0641 - 0.0576j, 0.0466 - 0.02275j],
          "h2": [0.0545 + 0.0165j, -1.3449 - 0.4523j, 1.0067 + 1.1524j, 0.3476 + 0.3153j]}[channel]
    ir = np.array(ir).astype(np.complex64)
    h_channel = np.zeros(sps * (len(ir) - 1) + 1, dtype=np.complex64)
    h_channel[0::sps] = ir
    h_channel /= np.linalg.norm(h_channel)
    nlev = {"4-QAM": 2, "16-QAM": 4, "64-QAM": 8}[mod]
    ask = np.arange(-(nlev - 1), nlev, 2).astype(np.float64)
    constellation = (ask[:, None] + 1j * ask[None, :]).reshape(-1)
    constellation = constellation / np.sqrt(np.mean(np.abs(constellation) ** 2))        # NN:233
    amp_levels = torch.tensor(constellation.real[::nlev], dtype=torch.float32)          # NN:235-237
    with SeededRng(seed):
        rx, data = nnref.generate_data(B * n_steps, len(ir), constellation, SNR, h_channel, sps, "cpu")
    net = nnref.Net(k1, k2, nlev, sps)
    h_est = np.zeros([2, M_est])
    h_est[0, M_est // 2] = 1
    h_est = torch.tensor(h_est, requires_grad=True, dtype=torch.float32)
    opt = torch.optim.Adam(net.parameters(), lr=lr, amsgrad=True)
    opt.add_param_group({"params": h_est})
    params = [net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias, h_est]
    flat = lambda ts: np.concatenate([t2n(t).reshape(-1) for t in ts])
    res = dict(rx=t2n(rx), data=t2n(data), theta0=flat(params), amp_levels=t2n(amp_levels), lr=np.float64(lr), B=np.int64(B),
               M_est=np.int64(M_est), k1=np.int64(k1), k2=np.int64(k2), sps=np.int64(sps), n_steps=np.int64(n_steps), mod=np.array(mod),
               SNR=np.float64(SNR))
    losses = np.zeros(n_steps, dtype=np.float32)
    mb = torch.empty(1, 2, B * sps)
    for s in range(n_steps):
        mb[0] = rx[:, s * B * sps:(s + 1) * B * sps]
        opt.zero_grad()
        q = net(mb)
        loss = nnref.loss_function(q.squeeze(), mb.squeeze(), h_est, "cpu", amp_levels)
        loss.backward()
        losses[s] = loss.item()
        if s < 3:
            res[f"q{s}"], res[f"g{s}"] = t2n(q[0]), flat([p_.grad for p_ in params])
        opt.step()
        if s < 3 or s + 1 == n_steps:
            res[f"theta{s + 1}"] = flat(params)
    res["loss"] = losses
    res["m"] = flat([opt.state[p_]["exp_avg"] for p_ in params])
    res["v"] = flat([opt.state[p_]["exp_avg_sq"] for p_ in params])
    res["vmax"] = flat([opt.state[p_]["max_exp_avg_sq"] for p_ in params])
    return res


def capture_G8(sfun, awgn):
    import contextlib
    import io
    import func_VAENN_MQAM as nnref  # reference module

    # the sweep script's shape (Eval_run_vaenn.py: 64-QAM, k1 = 25, k2 = 3, M = 25, batch_len 300, lr 4e-3, SNR 24)
    save("G8_vaenn_64qam", **_nn_case("64-QAM", 24, 25, 25, 3, 300, seed=81, n_steps=10))
    save("G8_vaenn_16qam_small", **_nn_case("16-QAM", 20, 9, 11, 3, 60, seed=82, n_steps=3, lr=2e-3))
    save("G8_vaenn_4qam_k5", **_nn_case("4-QAM", 12, 13, 7, 5, 41, seed=83, n_steps=3, channel="h2"))
    # processing() itself: 16-QAM, 120 epochs x 4 minibatches of 300, validation on 5000 symbols every 2nd epoch; the initial
    # parameters are what Net() draws right after torch.manual_seed(84), captured by constructing the same net first
    torch.manual_seed(84)
    net0 = nnref.Net(25, 3, 4, 2)
    h0 = np.zeros([2, 25])
    h0[0, 12] = 1
    theta0 = np.concatenate([t2n(p_).reshape(-1) for p_ in (net0.fc1.weight, net0.fc1.bias, net0.fc2.weight, net0.fc2.bias)] + [h0.reshape(-1)])
    t0 = time.time()
    torch.manual_seed(84)
    with SeededRng(84), contextlib.redirect_stdout(io.StringIO()):
        SER = nnref.processing("16-QAM", 2, 20, 25, 25, 3, 4e-3, 300, 5000, 1200, 120, 2, "h1", "Net")
    print(f"   G8 run: {time.time() - t0:.0f}s  SER {np.round(t2n(SER), 4).tolist()}")
    save("G8_vaenn_run", SER=t2n(SER), theta0=theta0.astype(np.float32), seed=np.int64(84), seconds=np.float64(time.time() - t0),
         args=np.array([str(v) for v in ("16-QAM", 2, 20, 25, 25, 3, 4e-3, 300, 5000, 1200, 120, 2, "h1", "Net")]))


def capture_G9(sfun, awgn):
    """Config 4 (optical DP 64-QAM VAEflex) through the reference's processing(): 70 frames x 2000 symbols = 190 window steps per
    frame, long enough to converge (~15 min on one core)."""
    import contextlib
    import io
    import func_VAEflex_DP_MQAM_shaping as ref_flex

    t0 = time.time()
    with SeededRng(91), contextlib.redirect_stdout(io.StringIO()):
        SER, Var_est, var = ref_flex.processing("64-QAM", 2, 23, 0.0, 25, 0.006 * np.pi, np.pi / 10, 2.5e-3, 100, 2000, 70, 10, "h0", 90e9,
                                                -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], dtype=np.complex64), 170)
    print(f"   G9 flex run: {time.time() - t0:.0f}s  SER last {SER[:, -1].tolist()}")
    save("G9_flex_run", SER=t2n(SER), Var_est=t2n(Var_est), var=t2n(var), seed=np.int64(91), seconds=np.float64(time.time() - t0),
         N_frame_max=np.int64(2000), num_frames=np.int64(70), theta_diff=np.float64(0.006 * np.pi))


def capture_G10(sfun, awgn):
    """Config 5's shape (optical DP 64-QAM + PCS, H = 5.72 bit): one VAE-LE run through the reference's processing(), 200 frames x 3000
    symbols at SNR 23 dB."""
    import contextlib
    import io
    import func_VAELE_DP_MQAM_shaping as ref_vaele

    t0 = time.time()
    with SeededRng(101), contextlib.redirect_stdout(io.StringIO()):
        SER, Var_est, var = ref_vaele.processing("64-QAM", 2, 23, NU_572, 25, 0.006 * np.pi, np.pi / 10, 2.5e-3, 100, 3000, 200, 10, "h0", 90e9,
                                                 -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], dtype=np.complex64), 170)
    print(f"   G10 PCS run: {time.time() - t0:.0f}s  SER last {SER[:, -1].tolist()}")
    save("G10_pcs_run", SER=t2n(SER), Var_est=t2n(Var_est), var=t2n(var), seed=np.int64(101), seconds=np.float64(time.time() - t0),
         N_frame_max=np.int64(3000), num_frames=np.int64(200), theta_diff=np.float64(0.006 * np.pi), nu=np.float64(NU_572))


def _nnbn_case(mod, SNR, M_est, k1, k2, B, seed, n_steps, lr=4e-3, channel="h1"):
    """Net_BN (NN:190-211) teacher-forced + free run, incl. the BatchNorm running statistics and an eval-mode forward."""
    import func_VAENN_MQAM as nnref  # reference module

    sps = 2
    torch.manual_seed(seed)
    ir = np.array({"h1": [0.0545 + 0.05j, 0.2823 - 0.11971j, -0.7676 + 0.2788j, -0.0641 - 0.0576j, 0.0466 - 0.02275j]}[channel]).astype(np.complex64)
    h_channel = np.zeros(sps * (len(ir) - 1) + 1, dtype=np.complex64)
    h_channel[0::sps] = ir
    h_channel /= np.linalg.norm(h_channel)
    nlev = {"4-QAM": 2, "16-QAM": 4, "64-QAM": 8}[mod]
    ask = np.arange(-(nlev - 1), nlev, 2).astype(np.float64)
    constellation = (ask[:, None] + 1j * ask[None, :]).reshape(-1)
    constellation = constellation / np.sqrt(np.mean(np.abs(constellation) ** 2))
    amp_levels = torch.tensor(constellation.real[::nlev], dtype=torch.float32)
    with SeededRng(seed):
        rx, data = nnref.generate_data(B * n_steps, len(ir), constellation, SNR, h_channel, sps, "cpu")
    net = nnref.Net_BN(k1, k2, nlev, sps)
    net.train()
    h_est = np.zeros([2, M_est])
    h_est[0, M_est // 2] = 1
    h_est = torch.tensor(h_est, requires_grad=True, dtype=torch.float32)
    opt = torch.optim.Adam(net.parameters(), lr=lr, amsgrad=True)
    opt.add_param_group({"params": h_est})
    params = [net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias, net.batch1.weight, net.batch1.bias, h_est]
    assert [id(p_) for p_ in net.parameters()] == [id(p_) for p_ in params[:-1]]          # the flat order used by the kernels
    flat = lambda ts: np.concatenate([t2n(t).reshape(-1) for t in ts])
    bnstat = lambda: np.concatenate([t2n(net.batch1.running_mean), t2n(net.batch1.running_var)])
    res = dict(rx=t2n(rx), theta0=flat(params), bn0=bnstat(), amp_levels=t2n(amp_levels), lr=np.float64(lr), B=np.int64(B), M_est=np.int64(M_est),
               k1=np.int64(k1), k2=np.int64(k2), sps=np.int64(sps), n_steps=np.int64(n_steps), mod=np.array(mod), SNR=np.float64(SNR))
    losses = np.zeros(n_steps, dtype=np.float32)
    mb = torch.empty(1, 2, B * sps)
    for s in range(n_steps):
        mb[0] = rx[:, s * B * sps:(s + 1) * B * sps]
        opt.zero_grad()
        q = net(mb)
        loss = nnref.loss_function(q.squeeze(), mb.squeeze(), h_est, "cpu", amp_levels)
        loss.backward()
        losses[s] = loss.item()
        if s < 2:
            res[f"q{s}"], res[f"g{s}"], res[f"bn{s + 1}"] = t2n(q[0]), flat([p_.grad for p_ in params]), bnstat()
        opt.step()
        if s < 2 or s + 1 == n_steps:
            res[f"theta{s + 1}"] = flat(params)
    res["loss"], res[f"bn{n_steps}"] = losses, bnstat()
    net.eval()
    with torch.no_grad():
        mb2 = rx[None, :, :B * sps * min(n_steps, 3)]
        res["q_eval"] = t2n(net(mb2)[0])                                        # running statistics
    res["vmax"] = flat([opt.state[p_]["max_exp_avg_sq"] for p_ in params])
    return res


def capture_G11(sfun, awgn):
    import contextlib
    import io
    import func_VAENN_MQAM as nnref  # reference module

    save("G11_vaennbn_64qam", **_nnbn_case("64-QAM", 24, 25, 25, 3, 300, seed=111, n_steps=6))
    save("G11_vaennbn_16qam_small", **_nnbn_case("16-QAM", 20, 9, 11, 3, 60, seed=112, n_steps=3, lr=2e-3))
    torch.manual_seed(114)
    net0 = nnref.Net_BN(25, 3, 4, 2)
    h0 = np.zeros([2, 25])
    h0[0, 12] = 1
    theta0 = np.concatenate([t2n(p_).reshape(-1) for p_ in net0.parameters()] + [h0.reshape(-1)])
    t0 = time.time()
    torch.manual_seed(114)
    with SeededRng(114), contextlib.redirect_stdout(io.StringIO()):
        SER = nnref.processing("16-QAM", 2, 20, 25, 25, 3, 4e-3, 300, 5000, 1200, 120, 2, "h1", "Net_BN")
    print(f"   G11 run: {time.time() - t0:.0f}s  SER {np.round(t2n(SER), 4).tolist()}")
    save("G11_vaennbn_run", SER=t2n(SER), theta0=theta0.astype(np.float32), seed=np.int64(114), seconds=np.float64(time.time() - t0))


# --------------------------------------------------------------------------
# G12: row f4, constant-modulus baselines CMA / CMAbatch / CMAflex + CPE (shared_funcs.py:139-186, 341-488)
# --------------------------------------------------------------------------
def capture_G12(sfun, awgn):
    import contextlib
    import io
    import func_CMA_DP_MQAM_shaping as ref_cma
    import func_CMAbatch_DP_MQAM_shaping as ref_cmab
    import func_CMAflex_DP_MQAM_shaping as ref_cmaf

    sps, M_est, N = 2, 25, 700
    h_est, h_channel, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", "64-QAM", "cpu", 0.0, sps, M_est, 23)
    with SeededRng(121):
        rx, data, _ = sfun.generate_data_shaping(N, amps, 23, h_channel, P, pol, DP_DEFAULTS["symb_rate"], sps, DP_DEFAULTS["tau_cd"],
                                                 DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"], DP_DEFAULTS["theta"], "cpu")
    res = dict(rx=t2n(rx), data=t2n(data), h0=t2n(h_est), lr_cma=np.float64(1e-3), lr_cmabatch=np.float64(5e-5), lr_cmaflex=np.float64(5e-6),
               sps=np.int64(sps), M_est=np.int64(M_est), batchlen=np.int64(100),
               symb_step=np.int64(10), amp_levels=t2n(amp_levels), var=t2n(var), nu_sc=np.float64(nu_sc))
    with torch.no_grad():
        for tag, fn in (("cma", lambda h: sfun.CMA(rx.clone(), 1, h, 1e-3, sps, True)),
                        ("cmabatch", lambda h: sfun.CMAbatch(rx.clone(), 1, h, 5e-5, 100, sps, True)),
                        ("cmaflex", lambda h: sfun.CMAflex(rx.clone(), 1, h, 5e-6, 100, 10, sps, True))):
            out, h, e = fn(h_est.detach().clone())
            res[f"{tag}_out"], res[f"{tag}_h"], res[f"{tag}_e"] = t2n(out), t2n(h), t2n(e)
        # CPE on a phase-rotated noisy constellation with a slow drift (so that the unwrapping matters)
        g = torch.Generator().manual_seed(5)
        n = 3000
        lev = amp_levels[torch.randint(0, 8, (2, 2, n), generator=g)]
        phi = 0.9 + 2.2 * torch.arange(n) / n
        y = torch.stack([torch.stack([lev[p, 0] * torch.cos(phi) - lev[p, 1] * torch.sin(phi), lev[p, 1] * torch.cos(phi) + lev[p, 0] * torch.sin(phi)])
                         for p in range(2)]) + 0.02 * torch.randn(2, 2, n, generator=g)
        res["cpe_in"], res["cpe_out"] = t2n(y), t2n(sfun.CPE(y.clone()))
    save("G12_cma", **res)
    runs = {}
    # 4-QAM at 18 dB: the constant-modulus criterion converges within ~12 frames of 1500 symbols; step sizes per variant (the batch
    # forms sum the increments of 100 symbols, CMAflex applies such a sum every 10 symbols)
    for tag, mod_, seed, lr in (("cma", ref_cma, 122, 1e-3), ("cmabatch", ref_cmab, 123, 1e-4), ("cmaflex", ref_cmaf, 124, 1e-5)):
        t0 = time.time()
        with SeededRng(seed), contextlib.redirect_stdout(io.StringIO()):
            SER, Var_est, var_ = mod_.processing("4-QAM", 2, 18, 0.0, 25, 0.006 * np.pi, np.pi / 10, lr, 100, 1500, 24, 10, "h0", 90e9, -26e-24,
                                                 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], dtype=np.complex64), 170)
        print(f"   G12 {tag}: {time.time() - t0:.0f}s  SER last {SER[:, -1].tolist()}")
        runs[f"{tag}_SER"], runs[f"{tag}_seed"], runs[f"{tag}_lr"] = t2n(SER), np.int64(seed), np.float64(lr)
    save("G12_cma_runs", **runs)


# --------------------------------------------------------------------------
# G1b: teacher-forced DP steps at config 5's heavy shaping (64-QAM, nu = 0.0872449 / 0.1222578, Eval_run_DP.py:24) -- where the
#      log P terms of the KL are largest
# --------------------------------------------------------------------------
def capture_G1b(sfun, awgn):
    save("G1_dp_step_64qam_nu0872", **_dp_case(sfun, "64-QAM", 0.0872449, 20, 25, 100, seed=15))
    save("G1_dp_step_64qam_nu1222", **_dp_case(sfun, "64-QAM", 0.1222578, 28, 25, 100, seed=16))


# --------------------------------------------------------------------------
# G13: config 5 on-grid points through the reference's processing() (Eval_run_DP.py:24,34 sweep vectors)
# --------------------------------------------------------------------------
def _g13_point(tag, nu, SNR, seed, N_frame_max, num_frames, theta_diff):
    import contextlib
    import io
    import func_VAELE_DP_MQAM_shaping as ref_vaele

    t0 = time.time()
    with SeededRng(seed), contextlib.redirect_stdout(io.StringIO()):
        SER, Var_est, var = ref_vaele.processing("64-QAM", 2, SNR, nu, 25, theta_diff, np.pi / 10, 2.5e-3, 100, N_frame_max, num_frames, 10, "h0", 90e9,
                                                 -26e-24, 0.1e-12 * np.sqrt(1000), np.array([0.0314, 0.0314], dtype=np.complex64), 170)
    print(f"   G13 {tag}: {time.time() - t0:.0f}s  SER last {SER[:, -1].tolist()}", flush=True)
    save(f"G13_cfg5_{tag}", SER=t2n(SER), Var_est=t2n(Var_est), var=t2n(var), seed=np.int64(seed), seconds=np.float64(time.time() - t0),
         N_frame_max=np.int64(N_frame_max), num_frames=np.int64(num_frames), theta_diff=np.float64(theta_diff), nu=np.float64(nu), SNR=np.float64(SNR))


def capture_G13a(sfun, awgn):      # reduced frame size (G10's): same-frames run-level parity
    _g13_point("nu0872_snr20", 0.0872449, 20, 131, 3000, 200, 0.006 * np.pi)


def capture_G13b(sfun, awgn):
    _g13_point("nu1222_snr28", 0.1222578, 28, 132, 3000, 200, 0.006 * np.pi)


def capture_G13c(sfun, awgn):      # script-faithful size (170 frames x 10 000 symbols, 0.06 pi drift): anchors of the 300-run grid
    _g13_point("full_nu0872_snr20", 0.0872449, 20, 133, 10000, 170, 0.06 * np.pi)


def capture_G13d(sfun, awgn):
    _g13_point("full_nu1222_snr28", 0.1222578, 28, 134, 10000, 170, 0.06 * np.pi)


# --------------------------------------------------------------------------
# G14: the CMA modules' two-stage epilogue at 16- / 64-QAM (func_CMA_DP_MQAM_shaping.py:39-53): SER_constell_shaping rescales the
#      slice VIEW of out_const in place (shared_funcs.py:242), so soft_dec / SER_IQflip see the normalised constellation.
#      Hand-driven in processing()'s order so that the per-frame tensors can be pinned; the SER rows equal processing()'s.
# --------------------------------------------------------------------------
def _g14_case(sfun, mod, SNR, nu, lr, seed, num_frames=16, N=1500, M_est=25):
    sps, N_cut = 2, 10
    h_est, h_channel, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", mod, "cpu", nu, sps, M_est, SNR)
    theta, theta_diff = np.pi / 10, 0.006 * np.pi
    SER_valid = torch.empty(4, num_frames)
    shifts = np.zeros((num_frames, 2, 2), dtype=np.int64)
    rs = np.zeros((num_frames, 2), dtype=np.int64)
    keep = {}
    with SeededRng(seed), torch.no_grad():
        for frame in range(num_frames):
            rx, data, _ = sfun.generate_data_shaping(N, amps, SNR, h_channel, P, pol, DP_DEFAULTS["symb_rate"], sps, DP_DEFAULTS["tau_cd"],
                                                     DP_DEFAULTS["tau_pmd"], DP_DEFAULTS["phiIQ"], theta, "cpu")
            out_const, h_est, e = sfun.CMA(rx, 1, h_est, lr, sps, True)
            theta += theta_diff
            if frame == num_frames - 1:
                keep.update(cma_out=t2n(out_const), data=t2n(data))
            out_const = sfun.CPE(out_const[:, :, N_cut:-N_cut])
            data = data[:, :, N_cut:-N_cut]
            shift, r = sfun.find_shift_symb_full(out_const, data, 21)
            shifts[frame, 0], rs[frame, 0] = shift.numpy(), r
            out_const = out_const.roll(r, 0)
            out_const[0, :, :], out_const[1, :, :] = out_const[0, :, :].roll(int(-shift[0]), -1), out_const[1, :, :].roll(int(-shift[1]), -1)
            ms = torch.max(torch.abs(shift))
            SER_valid[:2, frame] = sfun.SER_constell_shaping(out_const[:, :, 11:-11 - ms], data[:, :, 11:-11 - ms], amp_levels, nu_sc, var)
            if frame == num_frames - 1:
                keep.update(out_const_after=t2n(out_const))              # aligned; kept window rescaled in place by the call above
            out_train = sfun.soft_dec(out_const, var, amp_levels, nu_sc)
            shift, r = sfun.find_shift(out_train, data, 21, amp_levels, pol)
            shifts[frame, 1], rs[frame, 1] = shift.numpy(), r
            out_train = out_train.roll(r, 0)
            out_train[0, :, :], out_train[1, :, :] = out_train[0, :, :].roll(int(-shift[0]), -1), out_train[1, :, :].roll(int(-shift[1]), -1)
            ms = torch.max(torch.abs(shift))
            SER_valid[2:, frame] = sfun.SER_IQflip(out_train[:, :, 11:-11 - ms], data[:, :, 11:-11 - ms])
            print(f"   G14 {mod} frame {frame}: SER {np.round(SER_valid[:, frame].numpy(), 4).tolist()} shift {shifts[frame].tolist()} r {rs[frame].tolist()}",
                  flush=True)
    return dict(SER=t2n(SER_valid), shifts=shifts, rs=rs, amp_levels=t2n(amp_levels), var=t2n(var), nu_sc=np.float64(nu_sc), seed=np.int64(seed),
                lr=np.float64(lr), SNR=np.float64(SNR), nu=np.float64(nu), mod=np.array(mod), num_frames=np.int64(num_frames), N=np.int64(N),
                theta_diff=np.float64(theta_diff), **keep)


def capture_G13e(sfun, awgn):      # two converging anchors of the grid at the script-faithful size
    _g13_point("full_nu0271_snr26", NU_572, 26, 135, 10000, 170, 0.06 * np.pi)


def capture_G13f(sfun, awgn):
    _g13_point("full_nu0_snr20", 0.0, 20, 136, 10000, 170, 0.06 * np.pi)


def capture_G14(sfun, awgn):
    save("G14_cma_epilogue_64qam", **_g14_case(sfun, "64-QAM", 25, 0.0, 1e-3, seed=141))
    save("G14_cma_epilogue_16qam", **_g14_case(sfun, "16-QAM", 20, 0.0, 1e-3, seed=142))
    save("G14_cma_epilogue_64qam_pcs", **_g14_case(sfun, "64-QAM", 25, NU_572, 1e-3, seed=143, num_frames=12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--full-run", action="store_true", help="also capture the 170-frame default run (~13 min)")
    args = ap.parse_args()
    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    sfun, awgn = _import_reference()
    todo = [s for s in args.only.split(",") if s] or ["G0", "G1", "G2", "G3", "G4", "G5", "G6", "G7", "G8", "G9", "G10", "G11", "G12", "G1b", "G13a", "G13b", "G13c", "G13d", "G13e", "G13f", "G14"]
    for g in todo:
        print(f"[{g}]")
        if g == "G7":
            capture_G7(sfun, awgn, full=args.full_run)
        else:
            globals()[f"capture_{g}"](sfun, awgn)


if __name__ == "__main__":
    main()
