"""Host-side mirror of optical_DP_channel/shared_funcs.py for the VAE path (same names, argument order and
return shapes), backed by the HIP library.

What runs where:
  * ``init``, ``generate_data_shaping``            -- host (numpy), like the reference;
  * ``twoXtwoFIR.forward``, ``soft_dec``           -- HIP kernels (vaeq_dp_forward / vaeq_soft_demap);
  * ``loss_function_shaping``                      -- HIP (the fused step kernel run with ``no_update``);
  * ``find_shift*``, ``SER_*``                     -- torch ops on the tensors' device (epilogue.py);
  * training itself                                -- ``engine.DPEngine`` (the fused kernel); a
    ``net(...); loss.backward(); optimizer.step()`` loop written against these mirrors is NOT how the product trains.
"""
import numpy as np
import torch
import torch.nn as nn

from . import engine as _engine
from . import epilogue as _epi
from .channel import generate_data_shaping, rcfir, rrcfir, simulate_channel, simulate_dispersion  # noqa: F401

_CHANNELS = {  # impulse responses h1/h2 of Caciularu et al.; h0 = optical channel only (shared_funcs.py:545-550)
    "h1": [0.0545 + 0.05j, 0.2823 - 0.11971j, -0.7676 + 0.2788j, -0.0641 - 0.0576j, 0.0466 - 0.02275j],
    "h2": [0.0545 + 0.0165j, -1.3449 - 0.4523j, 1.0067 + 1.1524j, 0.3476 + 0.3153j],
    "h0": [1],
}
_LEVELS = {"4-QAM": 2, "16-QAM": 4, "64-QAM": 8}


def qam_tables(mod, nu):
    """Normalised square-QAM constellation, its ASK levels and the PCS pmf (shared_funcs.py:556-579)."""
    n = _LEVELS[mod]                                        # KeyError for an unknown format, like the reference (:563)
    ask = np.arange(-(n - 1), n, 2).astype(np.float64)      # -(n-1), ..., n-1
    const = (ask[:, None] + 1j * ask[None, :]).reshape(-1)  # row-major: real part slow, imaginary part fast (:556-559)
    const = const / np.sqrt(np.mean(np.abs(const) ** 2))
    amps = const.real[::n]
    sc = np.min(np.abs(amps))
    P = np.exp(-nu * np.abs(amps / sc) ** 2)
    P = P / np.sum(P)
    PP = np.tile(P, (n, 1))
    P_mat = (PP * PP.T) / np.sum(PP * PP.T)
    pow_mean = np.sum(P_mat.reshape(-1) * np.abs(const) ** 2)
    return dict(n=n, constellation=const, amps=amps, sc=sc, nu_sc=nu / sc ** 2, P=P, P_mat=P_mat, pow_mean=pow_mean)


def upsampled_channel(channel, sps):
    """Zero-stuffed, unit-norm extra impulse response (shared_funcs.py:552-554)."""
    if channel not in _CHANNELS:
        raise UnboundLocalError(f"unknown channel {channel!r} (the reference leaves h_channel_orig unbound, shared_funcs.py:545-552)")
    ir = np.array(_CHANNELS[channel]).astype(np.complex64)
    up = np.zeros(sps * (ir.shape[-1] - 1) + 1, dtype=np.complex64)
    up[0::sps] = ir
    return up / np.linalg.norm(up)


def init(channel, mod, device, nu, sps, M_est, SNR):
    """Constants of one run (shared_funcs.py:544-588) -> (h_est, h_channel, P, amp_levels, amps, pol, nu_sc, var, pow_mean)."""
    h_channel = upsampled_channel(channel, sps)
    t = qam_tables(mod, nu)
    pol = 2
    amp_levels = torch.tensor(t["amps"], device=device, dtype=torch.float32)
    var = torch.full((2,), t["pow_mean"] / 10 ** (SNR / 10) / 2, device=device, dtype=torch.float32)
    h0 = np.zeros([pol, pol, 2, M_est])
    h0[0, 0, 0, M_est // 2] = h0[1, 1, 0, M_est // 2] = 1
    h_est = torch.tensor(h0, requires_grad=True, dtype=torch.float32, device=device)
    return h_est, h_channel, t["P"], amp_levels, t["amps"], pol, t["nu_sc"], var, t["pow_mean"]


class twoXtwoFIR(nn.Module):
    """Complex 2x2 butterfly FIR + per-axis soft demapper (shared_funcs.py:490-527).

    ``conv_w.weight`` keeps the reference's Conv1d(4,2,M) layout so checkpoints/state_dicts interchange.
    forward() runs the HIP kernel (inference: no autograd graph is built; training goes through engine.DPEngine)."""

    def __init__(self, M_est, sps):
        super().__init__()
        self.sps = sps
        self.conv_w = nn.Conv1d(4, 2, M_est, bias=False, padding=M_est // 2, stride=sps).to(dtype=torch.float32)
        nn.init.dirac_(self.conv_w.weight)

    def forward(self, x, amp_levels, var, nu_sc):
        if torch.is_grad_enabled() and self.conv_w.weight.requires_grad:
            from .autograd_ops import fir_demap                  # HIP forward + HIP backward (vaeq_dp_forward / _bwd)
            return fir_demap(x, self.conv_w.weight, amp_levels, var, nu_sc, self.sps)
        q, y = _engine.dp_forward(x, self.conv_w.weight.detach(), amp_levels, var, nu_sc, self.sps)
        return q, y


def soft_dec(out, var, amp_levels, nu_sc):
    """Stand-alone soft demapper (shared_funcs.py:529-542): out[2,2,N] -> q[2,2n,N]."""
    return _engine.soft_demap(out, amp_levels, var, nu_sc)


def loss_function_shaping(q, rx, h_est, amp_levels, P):
    """ELBO of one minibatch (shared_funcs.py:92-137) -> (loss, var_est[2]); q[2,2n,B], rx[2,2,B*sps], h_est[2,2,2,M].

    HIP kernels: vaeq_dp_loss (values) and, when q or h_est require grad, vaeq_dp_loss_bwd through autograd_ops.elbo_loss."""
    if torch.is_grad_enabled() and (q.requires_grad or h_est.requires_grad):
        from .autograd_ops import elbo_loss
        return elbo_loss(q, rx, h_est, amp_levels, P)
    return _engine.dp_loss(q, rx, h_est.detach(), amp_levels, P)


# ------------------------------------------------------------------ row f4: constant-modulus baselines, reference signatures
def _cma(mode, Rx, R, h, lr, batchlen, symb_step, sps, eval):
    if not eval:
        lr = 0.0                                                # the reference skips the update (:370); outputs are the same
    hh = h.detach().reshape(1, 2, 2, 2, -1).contiguous().clone()
    out, e = _engine.cma(Rx.reshape(1, 2, 2, -1), hh, lr, sps, mode, batchlen, symb_step, float(R))
    with torch.no_grad():
        h.copy_(hh[0])                                          # the reference updates h in place, too
    return out[0], h, e[0]


def CMA(Rx, R, h, lr, sps, eval):
    """shared_funcs.py:341-383 on the device (vaeq_cma) -> (out[2,2,N//sps], h, e[N//sps,2])."""
    return _cma("CMA", Rx, R, h, lr, 100, 10, sps, eval)


def CMAbatch(Rx, R, h, lr, batchlen, sps, eval):
    """shared_funcs.py:385-433."""
    return _cma("CMAbatch", Rx, R, h, lr, batchlen, batchlen, sps, eval)


def CMAflex(Rx, R, h, lr, batchlen, symb_step, sps, eval):
    """shared_funcs.py:435-488."""
    return _cma("CMAflex", Rx, R, h, lr, batchlen, symb_step, sps, eval)


def CPE(y):
    """Viterbi-Viterbi carrier phase estimation (shared_funcs.py:139-186) on the device (vaeq_cpe)."""
    return _engine.cpe(y)


# ------------------------------------------------------------------ per-frame epilogue, reference signatures
def find_shift(q, tx, N_shift, amp_levels, pol):
    """shared_funcs.py:290-314 -> (shift[2] int16, r)."""
    n = q.shape[1] // 2
    E = torch.einsum("i,pin->pn", amp_levels, q[:, :n, :])
    shift, r = _epi.shift_search(E.unsqueeze(0), tx.unsqueeze(0), N_shift)
    return shift[0].to(torch.int16), int(r[0])


def find_shift_symb_full(rx, tx, N_shift):
    """shared_funcs.py:316-338."""
    shift, r = _epi.shift_search(rx[:, 0, :].unsqueeze(0), tx.unsqueeze(0), N_shift)
    return shift[0].to(torch.int16), int(r[0])


def SER_IQflip(q, tx):
    """shared_funcs.py:188-222 -> SER[2]."""
    n = q.shape[1] // 2
    dec = torch.stack([q[:, :n].argmax(dim=1), q[:, n:].argmax(dim=1)], dim=1).unsqueeze(0)
    mask = torch.ones(1, q.shape[-1], dtype=torch.bool, device=q.device)
    return _epi.ser_soft_demap(dec, tx.unsqueeze(0), mask, n)[0]


def SER_constell_shaping(rx, tx, amp_levels, nu_sc, var):
    """shared_funcs.py:225-265 -> SER[2].  Like the reference it rescales ``rx`` in place (:242)."""
    mask = torch.ones(1, rx.shape[-1], dtype=torch.bool, device=rx.device)
    txf = tx.float()
    rx *= torch.mean(torch.sqrt(txf[:, 0] ** 2 + txf[:, 1] ** 2)) / torch.mean(torch.sqrt(rx[:, 0] ** 2 + rx[:, 1] ** 2))
    nu = torch.as_tensor([nu_sc], dtype=torch.float32, device=rx.device)
    return _epi.ser_constellation(rx.unsqueeze(0), tx.unsqueeze(0), mask, amp_levels, nu, var[:1].reshape(1))[0]
