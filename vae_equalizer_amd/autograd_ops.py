"""torch.autograd.Function wrappers of the two stand-alone operators, forward AND backward on HIP kernels.

They make a reference-style training loop

    q, out = net(minibatch, amp_levels, var, nu_sc)                       # shared_funcs.twoXtwoFIR
    loss, var_est = loss_function_shaping(q, minibatch, h_est, amp_levels, P)
    loss.backward(); optimizer.step()                                     # func_VAELE_DP_MQAM_shaping.py:60-66

work on the GPU with torch.optim -- the differentiable form of the drop-in operator surface.  It is NOT how
the product trains (one fused kernel does forward, loss, backward and Adam: engine.DPEngine); it exists so that code written
against the reference's operators keeps working.  No CPU path: CPU tensors are refused by _native.ptr().
"""
import torch

from . import _native as nat
from .engine import _f32, dp_forward, dp_loss


class _FIRDemap(torch.autograd.Function):
    """twoXtwoFIR.forward (shared_funcs.py:500-527): (x[2,2,L], W[2,4,M]) -> (q[2,2n,B], out[2,2,B])."""

    @staticmethod
    def forward(ctx, x, W, amp, var, nu_sc, sps):
        q, y = dp_forward(x, W, amp, var, nu_sc, sps)
        ctx.save_for_backward(x, q, y, amp, var)
        ctx.sps, ctx.M = sps, W.shape[-1]
        return q, y

    @staticmethod
    def backward(ctx, gq, gy):
        x, q, y, amp, var = ctx.saved_tensors
        gq = gq.contiguous()
        gy = gy.contiguous() if gy is not None else None
        return None, _fir_bwd(x, q, y, gq, gy, amp, var, ctx.sps, ctx.M), None, None, None, None


def _fir_bwd(x, q, y, gq, gy, amp, var, sps, M):
    dev, N = x.device, q.shape[-1]
    n = amp.numel()
    gW = torch.empty(2, 4, M, dtype=torch.float32, device=dev)
    var2 = _f32(var, dev).reshape(1, 2).contiguous()
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_dp_forward_bwd(1, N, sps, M, n, nat.ptr(x.contiguous()), nat.ptr(q.contiguous()), nat.ptr(y.contiguous()),
                                                nat.ptr(gq), nat.ptr(gy), nat.ptr(amp), nat.ptr(var2), nat.ptr(gW),
                                                nat.current_stream(dev)), "vaeq_dp_forward_bwd")
    return gW


def fir_demap(x, W, amp_levels, var, nu_sc, sps):
    """Differentiable twoXtwoFIR.forward: gradients flow to W (x is data)."""
    amp = _f32(amp_levels, x.device).reshape(-1)
    var_t = _f32(var, x.device).reshape(2)
    return _FIRDemap.apply(x.contiguous(), W, amp, var_t, float(nu_sc), int(sps))


class _Loss(torch.autograd.Function):
    """loss_function_shaping (shared_funcs.py:92-137): (q, h_est) -> (loss, var_est); var_est is detached like the reference (:137)."""

    @staticmethod
    def forward(ctx, q, rx, h, amp, P):
        loss, ve = dp_loss(q, rx, h, amp, P)
        ctx.save_for_backward(q, rx, h, amp, P)
        ctx.mark_non_differentiable(ve)
        return loss, ve

    @staticmethod
    def backward(ctx, g_loss, g_ve):
        q, rx, h, amp, P = ctx.saved_tensors
        dev, B = q.device, q.shape[-1]
        sps, M, n = rx.shape[-1] // B, h.shape[-1], amp.numel()
        gq = torch.empty_like(q)
        gh = torch.empty_like(h)
        up = g_loss.reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            nat.check(nat.lib().vaeq_dp_loss_bwd(1, B, sps, M, n, nat.ptr(q.contiguous()), nat.ptr(rx.contiguous()), nat.ptr(h.contiguous()),
                                                 nat.ptr(amp), nat.ptr(P), nat.ptr(up), nat.ptr(gq), nat.ptr(gh),
                                                 nat.current_stream(dev)), "vaeq_dp_loss_bwd")
        return gq, None, gh, None, None


def elbo_loss(q, rx, h_est, amp_levels, P):
    """Differentiable loss_function_shaping: gradients flow to q and h_est."""
    dev = q.device
    amp = _f32(amp_levels, dev).reshape(-1)
    Pt = _f32(P, dev).reshape(1, -1).contiguous()
    return _Loss.apply(q, rx.contiguous(), h_est, amp, Pt)


# ------------------------------------------------------------------ the single-polarisation (AWGN) pair
class _AwgnFIRDemap(torch.autograd.Function):
    """twoFIR.forward (AWGN_channel/func_VAELE_MQAM_shaping.py:214-231): (x[2,L], W[1,2,M]) -> (q[2n,B], out[2,B])."""

    @staticmethod
    def forward(ctx, x, W, amp, amp_mean, var, sps):
        from .engine import AWGNEngine
        M = W.shape[-1]
        eng = AWGNEngine(1, M, amp, torch.full((amp.numel(),), 1.0 / amp.numel()), float(amp_mean), float(var), x.device, sps)
        eng.set_state(W.detach(), None)
        q, y = eng.forward(x.reshape(1, 2, -1))
        ctx.save_for_backward(x, W.detach().reshape(1, 2, M).contiguous(), amp, eng.amp_mean, eng.var)
        ctx.sps, ctx.wshape = sps, tuple(W.shape)
        return q[0], y[0]

    @staticmethod
    def backward(ctx, gq, gy):
        x, W, amp, amp_mean, var = ctx.saved_tensors
        dev, N, M = x.device, gq.shape[-1], W.shape[-1]
        gW = torch.empty(1, 2, M, dtype=torch.float32, device=dev)
        gy = gy.contiguous() if gy is not None else None
        with torch.cuda.device(dev):
            nat.check(nat.lib().vaeq_awgn_forward_bwd(1, N, ctx.sps, M, amp.numel(), nat.ptr(x.contiguous()), nat.ptr(W), nat.ptr(amp),
                                                      nat.ptr(amp_mean), nat.ptr(var), nat.ptr(gq.contiguous()), nat.ptr(gy), nat.ptr(gW),
                                                      nat.current_stream(dev)), "vaeq_awgn_forward_bwd")
        return None, gW.reshape(ctx.wshape), None, None, None, None


def awgn_fir_demap(x, W, amp_levels, amp_mean, var, sps):
    """Differentiable twoFIR.forward: gradients flow to W (x is data)."""
    return _AwgnFIRDemap.apply(x.contiguous(), W, _f32(amp_levels, x.device).reshape(-1), float(amp_mean), float(var), int(sps))


class _AwgnLoss(torch.autograd.Function):
    """loss_function of the AWGN modules (func_VAELE_MQAM_shaping.py:63-95 with the prior P, func_VAENN_MQAM.py:63-95 without):
    (q[2n,B], h[2,M]) -> loss."""

    @staticmethod
    def forward(ctx, q, rx, h, amp, P):
        from .engine import awgn_loss
        ctx.save_for_backward(q.detach(), rx, h.detach(), amp, P if P is not None else torch.empty(0, device=q.device))
        ctx.has_P = P is not None
        return awgn_loss(q.detach(), rx, h.detach(), amp, P)

    @staticmethod
    def backward(ctx, g):
        q, rx, h, amp, P = ctx.saved_tensors
        dev, B, M, n = q.device, q.shape[-1], h.shape[-1], amp.numel()
        gq = torch.empty(1, 2 * n, B, dtype=torch.float32, device=dev)
        gh = torch.empty(1, 2, M, dtype=torch.float32, device=dev)
        Pt = P.reshape(1, n).contiguous() if ctx.has_P else None
        up = g.reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            nat.check(nat.lib().vaeq_awgn_loss_bwd(1, B, rx.shape[-1] // B, M, n, nat.ptr(q.contiguous()), nat.ptr(rx.contiguous()),
                                                   nat.ptr(h.contiguous()), nat.ptr(amp), nat.ptr(Pt), nat.ptr(up), nat.ptr(gq), nat.ptr(gh),
                                                   nat.current_stream(dev)), "vaeq_awgn_loss_bwd")
        return gq[0], None, gh[0], None, None


def awgn_elbo_loss(q, rx, h_est, amp_levels, P=None):
    """Differentiable AWGN loss_function: gradients flow to q and h_est.  P None = the VAE-NN form (entropy instead of KL)."""
    dev = q.device
    Pt = None if P is None else _f32(P, dev).reshape(-1).contiguous()
    return _AwgnLoss.apply(q, rx.contiguous().float(), h_est, _f32(amp_levels, dev).reshape(-1).contiguous(), Pt)
