"""Drop-in for AWGN_channel/func_VAELE_MQAM_shaping.py: same ``processing`` signature (:235) and return value (:324),
with the training loop and the validation forward pass on the HIP kernels (engine.AWGNEngine)."""
import numpy as np
import torch

from . import channel as ch
from .dp_runs import _host_pool, default_device, resolve_generator
from .engine import AWGNEngine
from .shared_funcs import _CHANNELS, qam_tables


def awgn_tables(mod, nu, SNR, channel, sps):
    """Constants processing() derives before the loop (:239-272)."""
    if channel not in ("h1", "h2"):
        raise UnboundLocalError(f"unknown channel {channel!r} (the reference leaves h_channel_orig unbound, :239-244)")
    ir = np.array(_CHANNELS[channel]).astype(np.complex64)
    h_channel = np.zeros(sps * (ir.shape[-1] - 1) + 1, dtype=np.complex64)
    h_channel[0::sps] = ir
    h_channel /= np.linalg.norm(h_channel)
    t = qam_tables(mod, nu)
    n = t["n"]
    PP = np.tile(t["P"], (n, 1))
    shaped = (PP * PP.T).reshape(-1) * t["constellation"]                       # :270
    amp_mean = np.sum(np.abs(shaped.real) + np.abs(shaped.imag)) / 2             # :271
    return dict(amps=t["amps"], P=t["P"], amp_mean=amp_mean, var=10 ** (-SNR / 10), h_channel=h_channel, M_channel=len(ir), n=n)


generate_data = ch.generate_data                       # (:39-61) host restatement, reference signature + optional rng / noise streams


class twoFIR(torch.nn.Module):
    """Complex FIR + output normalisation + per-axis soft demapper (:206-231); ``conv_w.weight`` keeps the reference's
    Conv1d(2, 1, M) layout.  forward() runs vaeq_awgn_forward; with autograd enabled its backward is the HIP kernel vaeq_awgn_forward_bwd (the fused
    engine.AWGNEngine is how processing() trains)."""

    def __init__(self, M_est, sps):
        super().__init__()
        self.sps = sps
        self.conv_w = torch.nn.Conv1d(2, 1, M_est, bias=False, padding=M_est // 2, stride=sps).to(dtype=torch.float32)
        torch.nn.init.dirac_(self.conv_w.weight)

    def forward(self, x, amp_levels, amp_mean, var):
        if torch.is_grad_enabled() and self.conv_w.weight.requires_grad:
            from .autograd_ops import awgn_fir_demap              # HIP forward + HIP backward (vaeq_awgn_forward / _bwd)
            return awgn_fir_demap(x, self.conv_w.weight, amp_levels, amp_mean, var, self.sps)
        W = self.conv_w.weight.detach()
        eng = AWGNEngine(1, W.shape[-1], amp_levels, np.full(len(amp_levels), 1.0 / len(amp_levels)), float(amp_mean), float(var), x.device,
                         self.sps)
        eng.set_state(W, None)
        q, y = eng.forward(x.reshape(1, 2, -1))
        return q[0], y[0]


def loss_function(q, rx, h, device, amp_levels, P):
    """ELBO of one minibatch (:63-95): q[2n,B], rx[2,B*sps], h[2,M] (HIP: vaeq_awgn_loss; with autograd also vaeq_awgn_loss_bwd)."""
    if torch.is_grad_enabled() and (q.requires_grad or h.requires_grad):
        from .autograd_ops import awgn_elbo_loss
        return awgn_elbo_loss(q, rx, h, amp_levels, P)
    from .engine import awgn_loss
    return awgn_loss(q, rx, h.detach(), amp_levels, P)


def find_shift(q, tx, N_shift, amp_levels, num_lev, device=None):
    """:188-204 -- lag of the best |correlation| between E_q[x_I] and the TX I (or Q) sequence over the first 1000 symbols."""
    E = torch.einsum("i,in->n", amp_levels, q[:num_lev, :1000])
    half = N_shift // 2
    E_mat = torch.stack([torch.roll(E, i - half, 0) for i in range(N_shift)], dim=1)
    corr = tx[0, :1000].to(torch.float32) @ E_mat
    if torch.max(torch.abs(corr)) >= 0.02 * q.shape[-1]:
        return half - torch.argmax(torch.abs(corr))
    corr_IQ = tx[1, :1000].to(torch.float32) @ E_mat
    if torch.max(torch.abs(corr_IQ)) >= torch.max(torch.abs(corr)):
        return half - torch.argmax(torch.abs(corr_IQ))
    return half - torch.argmax(torch.abs(corr))


def SER_q(q, tx, sps, num_lev, device=None):
    """:97-123 -- SER of argmax(q) against TX, minimum over the four quadrant rotations."""
    N = tx.shape[-1]
    scale = (num_lev - 1) / 2
    data = torch.round(scale * tx.float() + scale)
    dec = torch.stack([q[:num_lev, :N].argmax(dim=0), q[num_lev:, :N].argmax(dim=0)]).float()
    dec_pi = -(dec - scale * 2)
    dec_pi4 = torch.stack([-(dec[1] - scale * 2), dec[0]])
    dec_3pi4 = -(dec_pi4 - scale * 2)
    return torch.stack([((data - d) != 0).any(dim=0).float().mean() for d in (dec, dec_pi, dec_pi4, dec_3pi4)]).min()


def SER_symb(rx, tx, sps, amp_levels, num_lev, device=None):
    """:125-153 -- SER of nearest-level decisions on the (un-equalised) symbol-rate samples, each axis normalised to unit power,
    minimum over the four quadrant rotations (the reference keeps it for its commented-out "unprocessed SER" print, :315)."""
    N = tx.shape[1]
    scale = (num_lev - 1) / 2
    data = torch.round(scale * tx.float() + scale)
    sI, sQ = rx[0, :N * sps:sps], rx[1, :N * sps:sps]
    sI, sQ = sI / torch.sqrt(2 * torch.mean(sI ** 2)), sQ / torch.sqrt(2 * torch.mean(sQ ** 2))
    dec = torch.stack([torch.argmin(torch.abs(sI[None] - amp_levels[:, None]), dim=0),
                       torch.argmin(torch.abs(sQ[None] - amp_levels[:, None]), dim=0)]).float()
    dec_pi = -(dec - scale * 2)
    dec_pi4 = torch.stack([-(dec[1] - scale * 2), dec[0]])
    dec_3pi4 = -(dec_pi4 - scale * 2)
    return torch.stack([((data - d) != 0).any(dim=0).float().mean() for d in (dec, dec_pi, dec_pi4, dec_3pi4)]).min()


def run_awgn_batch(runs, mod, sps, M_est, batch_len, N_valid, N_train, num_epochs, epe, channel, device=None, verbose=False,
                   generator=None, seed=None):
    """R AWGN VAE-LE runs at once: ``runs`` = list of dict(SNR, nu, lr_optim, seed).  Per epoch ONE training launch and, on evaluated
    epochs, ONE fused validation launch (forward + find_shift + SER_q, vaeq_awgn_validate) for all runs (:291-322).

    generator: None    = "hip" when no run carries a seed (the reference seeds nothing), else "numpy" (dp_runs.resolve_generator);
               "numpy" = the bit-faithful host channel model per run (seeded like tools/capture_golden.py when the run has a seed);
               "hip"   = the on-device generator (vaeq_gen_awgn), Philox streams keyed by ``seed``, the draw counter and the run index.
    Returns SER_valid[R, num_epochs // epe] (CPU float32)."""
    device = default_device() if device is None else torch.device(device)
    R = len(runs)
    generator = resolve_generator(generator, any(r.get("seed") is not None for r in runs))
    if seed is None:                                                             # Philox key of the device generator: fresh entropy when not given
        from .dp_runs import fresh_seed
        seed = fresh_seed()
    tab_of = {k: awgn_tables(mod, k[0], k[1], channel, sps) for k in {(r["nu"], r["SNR"]) for r in runs}}   # one table set per sweep point, not per run
    tabs = [tab_of[(r["nu"], r["SNR"])] for r in runs]
    t0 = tabs[0]
    amp = torch.tensor(t0["amps"], dtype=torch.float32, device=device)
    eng = AWGNEngine(R, M_est, amp, np.stack([t["P"] for t in tabs]), [t["amp_mean"] for t in tabs], [t["var"] for t in tabs],
                     device, sps)
    lr = np.array([r["lr_optim"] for r in runs], dtype=np.float32)
    streams = [ch.SeededStreams(r["seed"]) if r.get("seed") is not None else None for r in runs]
    steps = N_train // batch_len                                                 # :297 (the remainder is dropped)
    n_eval = num_epochs // epe
    SER_dev = torch.empty(R, max(n_eval, 1), dtype=torch.float32, device=device)
    P_all = np.stack([t["P"] for t in tabs])
    snr_all = np.array([r["SNR"] for r in runs], dtype=np.float32)
    draws = [0]

    def draw(N):
        if generator == "hip":
            draws[0] += 1
            return ch.generate_awgn_batch_hip(R, N, t0["amps"], P_all, snr_all, t0["h_channel"], sps, device, seed, draws[0] - 1)
        if generator != "numpy":
            raise ValueError(f"unknown generator {generator!r}")
        def host(i):
            t, r, st = tabs[i], runs[i], streams[i]
            return ch.generate_data(N, t["M_channel"], t["amps"], r["SNR"], t["h_channel"], sps, "cpu", t["P"],
                                    rng=st.next_rng() if st else None, noise=st.noise if st else None)
        seeded = R > 1 and all(st is not None for st in streams)                   # own random streams: safe to generate concurrently
        pairs = list(_host_pool().map(host, range(R))) if seeded else [host(i) for i in range(R)]
        return torch.stack([p[0] for p in pairs]).to(device), torch.stack([p[1] for p in pairs]).to(device)

    for epoch in range(num_epochs):
        rx, _ = draw(N_train)
        out = eng.train(rx, batch_len, steps, lr)
        if epoch % epe == 0 and epoch // epe < n_eval:                           # :308-318
            if generator == "hip" and ch.awgn_clean_supported(sps, M_est):       # the validation frame is read once: its noise goes on while it is read
                draws[0] += 1
                ser, sh, _ = eng.validate_clean(ch.generate_awgn_clean_batch_hip(R, N_valid, t0["amps"], P_all, snr_all, t0["h_channel"], sps,
                                                                                 device, seed, draws[0] - 1), 21)
            else:
                rxv, datav = draw(N_valid)
                ser, sh, _ = eng.validate(rxv, datav, 21)
            SER_dev[:, epoch // epe] = ser
            if verbose:
                loss, ser_h, sh_h = out["loss"][:, -1].cpu(), ser.cpu(), sh.cpu()
                for i in range(R):
                    tag = f"[run {i}] " if R > 1 else ""
                    print(f"{tag}{epoch}", loss[i].item(), int(sh_h[i]), '\t\t\t\t\t\tSER = ', ser_h[i].item())
    return SER_dev[:, :n_eval].cpu()


def processing(mod, sps, SNR, nu, M_est, lr_optim, batch_len, N_valid, N_train, num_epochs, epe, channel, *, seed=None,
               device=None, verbose=True, generator=None):
    """One AWGN VAE-LE run -> SER_valid[num_epochs//epe] (CPU float32).

    NB the sweep script passes its ``N_train`` (350) as ``batch_len`` and ``train_len`` (1200) as ``N_train``
    (Eval_run_shaping_vaele.py:53)."""
    device = default_device() if device is None else torch.device(device)
    if verbose:
        print("We are using the following device for learning:", device)
    return run_awgn_batch([dict(SNR=SNR, nu=nu, lr_optim=lr_optim, seed=seed)], mod, sps, M_est, batch_len, N_valid, N_train,
                          num_epochs, epe, channel, device=device, verbose=verbose, generator=generator)[0]
