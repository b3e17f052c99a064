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
