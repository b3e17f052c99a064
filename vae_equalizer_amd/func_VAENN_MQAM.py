"""Drop-in for AWGN_channel/func_VAENN_MQAM.py: same ``processing`` signature (:215) and return value (:304); the
training loop and the validation pass run on the HIP kernels (engine.NNEngine: vaeq_nn_train, vaeq_nn_validate).

Both topologies of the reference are implemented: ``net_type='Net'`` and ``'Net_BN'`` (BatchNorm1d between the ELU and fc2)."""
import numpy as np
import torch

from . import channel as ch
from .dp_runs import _host_pool, default_device
from .engine import NNEngine
from .func_VAELE_MQAM_shaping import SER_q, SER_symb, find_shift  # noqa: F401  (identical helpers in both reference files)
from .shared_funcs import _CHANNELS, _LEVELS


def vaenn_tables(mod, channel, sps):
    """Constants processing() derives before the loop (:219-237): unit-power square QAM, its ASK levels, the upsampled unit-norm
    channel impulse response."""
    if channel not in ("h1", "h2"):
        raise UnboundLocalError(f"unknown channel {channel!r} (the reference leaves h_channel_orig unbound, :219-222)")
    ir = np.array(_CHANNELS[channel]).astype(np.complex64)
    h_channel = np.zeros(sps * (ir.shape[-1] - 1) + 1, dtype=np.complex64)
    h_channel[0::sps] = ir
    h_channel /= np.linalg.norm(h_channel)
    n = _LEVELS[mod]
    ask = np.arange(-(n - 1), n, 2).astype(np.float64)
    constellation = (ask[:, None] + 1j * ask[None, :]).reshape(-1)
    constellation = constellation / np.sqrt(np.mean(np.abs(constellation) ** 2))
    return dict(constellation=constellation, amps=constellation.real[::n], h_channel=h_channel, M_channel=len(ir), n=n)


def generate_data(N, M, constellation, SNR, h_channel, sps, device, rng=None):
    """Host restatement of generate_data (:39-61): uniform symbols, RRC + channel, AWGN of FIXED variance 1/(2 SNR).
    rng: a ``np.random.RandomState`` (the reference uses the global legacy stream for symbols and noise alike)."""
    rng = np.random if rng is None else rng
    T = ch.PULSE_SPAN
    N_conv = N + len(h_channel) + 4 * T
    data = rng.randint(len(constellation), size=N_conv)
    tx_sig = constellation[data]
    tx_up = np.zeros(sps * (N_conv - 1) + 1, dtype=np.complex64)
    tx_up[::sps] = tx_sig
    sig = np.convolve(np.convolve(tx_up, ch.rrcfir(T, sps, ch.ROLL_OFF), mode="valid"), h_channel, mode="valid")
    sigma_n = np.sqrt(1 / 2) / 10 ** (SNR / 20)
    sig = sig + sigma_n * (rng.randn(*sig.shape) + 1j * rng.randn(*sig.shape))
    lo = T + M - 1
    rx = np.stack([sig[:sps * N].real, sig[:sps * N].imag])
    ref = np.stack([tx_sig[lo:lo + N].real, tx_sig[lo:lo + N].imag])
    return (torch.from_numpy(np.ascontiguousarray(rx)).to(device, torch.float32),
            torch.from_numpy(np.ascontiguousarray(ref)).to(device, torch.float16))


def loss_function(q, rx, h, device, amp_levels):
    """ELBO of one minibatch (:63-95): q[2n,B], rx[2,B*sps], h[2,M] (HIP: vaeq_awgn_loss with the entropy term; differentiable)."""
    if torch.is_grad_enabled() and (q.requires_grad or h.requires_grad):
        from .autograd_ops import awgn_elbo_loss
        return awgn_elbo_loss(q, rx, h, amp_levels, None)
    from .engine import awgn_loss
    return awgn_loss(q, rx, h.detach(), amp_levels, None)


def run_vaenn_batch(runs, mod, sps, M_est, kernel_1, kernel_2, batch_len, N_valid, N_train, num_epochs, epe, channel, device=None,
                    verbose=False, generator="hip", seed=0, theta0=None, net_type="Net"):
    """R VAE-NN runs at once: ``runs`` = list of dict(SNR, lr_optim, seed).  Per epoch one generator call, ONE training launch
    (N_train // batch_len minibatches) and, on evaluated epochs, one fused validation launch for all runs (:266-301).

    generator: "hip" = on-device channel model (vaeq_gen_awgn with the script's fixed noise level), Philox streams keyed by ``seed``;
               "numpy" = the host restatement per run, seeded per run when the run has a seed.
    theta0: optional [R, NP] initial parameters (default: Xavier / PyTorch-default initialisation drawn on the device).
    Returns SER_valid[R, num_epochs // epe] (CPU float32)."""
    device = default_device() if device is None else torch.device(device)
    R = len(runs)
    t = vaenn_tables(mod, channel, sps)
    if net_type not in ("Net", "Net_BN"):
        raise UnboundLocalError(f"unknown net_type {net_type!r} (the reference leaves `net` unbound, :239-243)")
    eng = NNEngine(R, M_est, kernel_1, kernel_2, t["amps"], device, sps, batch_norm=(net_type == "Net_BN"))
    if theta0 is None:
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed) + 12345)
        eng.init_parameters(gen)
    else:
        eng.init_parameters()
        eng.theta.copy_(torch.as_tensor(theta0, dtype=torch.float32, device=device).expand(R, -1))
    lr = np.array([r["lr_optim"] for r in runs], dtype=np.float32)
    snr = np.array([r["SNR"] for r in runs], dtype=np.float32)
    sigma = np.sqrt(0.5) / 10 ** (snr / 20)
    P = np.full(t["n"], 1.0 / t["n"])
    rngs = [np.random.RandomState(r["seed"]) if r.get("seed") is not None else None for r in runs]
    steps = N_train // batch_len
    n_eval = num_epochs // epe
    SER_dev = torch.empty(R, max(n_eval, 1), dtype=torch.float32, device=device)
    draws = [0]

    def draw(N):
        if generator == "hip":
            draws[0] += 1
            return ch.generate_awgn_batch_hip(R, N, t["amps"], P, snr, t["h_channel"], sps, device, seed, draws[0] - 1, sigma_fixed=sigma)
        if generator != "numpy":
            raise ValueError(f"unknown generator {generator!r}")
        host = lambda i: generate_data(N, t["M_channel"], t["constellation"], runs[i]["SNR"], t["h_channel"], sps, "cpu", rngs[i])
        seeded = R > 1 and all(g is not None for g in rngs)                         # own random streams: safe to generate concurrently
        pairs = list(_host_pool().map(host, range(R))) if seeded else [host(i) for i in range(R)]
        return torch.stack([p[0] for p in pairs]).to(device), torch.stack([p[1] for p in pairs]).to(device)

    for epoch in range(num_epochs):
        rx, _ = draw(N_train)
        out = eng.train(rx, batch_len, steps, lr)
        if epoch % epe == 0 and epoch // epe < n_eval:
            rxv, datav = draw(N_valid)
            ser, sh = eng.validate(rxv, datav, 21)
            SER_dev[:, epoch // epe] = ser
            if verbose:
                loss, ser_h, sh_h = out["loss"][:, -1].cpu(), ser.cpu(), sh.cpu()
                for i in range(R):
                    tag = f"[run {i}] " if R > 1 else ""
                    print(f"{tag}{epoch}", loss[i].item(), int(sh_h[i]), '\t\t\t\t\t\tSER = ', ser_h[i].item())
    return SER_dev[:, :n_eval].cpu()


def processing(mod, sps, SNR, M_est, kernel_1, kernel_2, lr_optim, batch_len, N_valid, N_train, num_epochs, epe, channel, net_type, *,
               seed=None, device=None, verbose=True, generator="numpy", theta0=None):
    """One VAE-NN run -> SER_valid[num_epochs//epe] (CPU float32), the reference's positional signature (:215)."""
    device = default_device() if device is None else torch.device(device)
    if verbose:
        print("We are using the following device for learning:", device)
    return run_vaenn_batch([dict(SNR=SNR, lr_optim=lr_optim, seed=seed)], mod, sps, M_est, kernel_1, kernel_2, batch_len, N_valid, N_train,
                           num_epochs, epe, channel, device=device, verbose=verbose, generator=generator, seed=seed or 0, theta0=theta0,
                           net_type=net_type)[0]
