"""Batched run engines: device-resident state of R independent equalizer runs + launches of the HIP training loop.

This is the host side of the C ABI (include/vaeq.h).  It mirrors what one call of the reference's
``processing()`` keeps alive between minibatches -- the ``twoXtwoFIR`` weight, ``h_est`` and the two
``optim.Adam`` parameter groups (optical_DP_channel/func_VAELE_DP_MQAM_shaping.py:26-31) -- for R runs at
once, because on an MI355X the sweep (Eval_run_DP.py:68-86), not the single run, is the unit of parallelism.
"""
import ctypes as C

import torch

from . import _native as nat


def _f32(x, device):
    return torch.as_tensor(x, dtype=torch.float32).to(device).contiguous()


class DPEngine:
    """R dual-polarisation runs (VAE-LE or VAEflex) sharing modulation, M_est and sps; P/var/nu_sc/lr per run."""

    def __init__(self, R, M_est, amp_levels, P, var, nu_sc, device="cuda:0", sps=2, threads=0):
        if M_est % 2 == 0:
            # even M_est: FIR yields B+1 outputs and the loss indexes M_est+1 taps in the reference (SURVEY note N3)
            raise ValueError("M_est must be odd")
        self.device = torch.device(device)
        self.R, self.M, self.sps, self.threads = int(R), int(M_est), int(sps), int(threads)
        self.amp = _f32(amp_levels, self.device).reshape(-1)
        self.n_lev = self.amp.numel()
        self.P = self._per_run(P, (self.n_lev,))
        self.var = self._per_run(var, (2,))
        self.nu_sc = self._per_run(nu_sc, ())
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.W, self.h = z(R, 2, 4, self.M), z(R, 2, 2, 2, self.M)
        self.mW, self.vW, self.mh, self.vh = z(R, 2, 4, self.M), z(R, 2, 4, self.M), z(R, 2, 2, 2, self.M), z(R, 2, 2, 2, self.M)
        self.step = torch.zeros(R, dtype=torch.int32, device=self.device)
        self.reset()

    def _per_run(self, x, shape):
        t = _f32(x, self.device)
        if t.dim() == len(shape):
            t = t.expand(self.R, *shape)
        if tuple(t.shape) != (self.R, *shape):
            raise ValueError(f"expected shape {(self.R, *shape)} or {shape}, got {tuple(t.shape)}")
        return t.contiguous()

    def reset(self):
        """Dirac initialisation of W (shared_funcs.py:495) and h_est (:585); Adam state zeroed."""
        for t in (self.W, self.h, self.mW, self.vW, self.mh, self.vh):
            t.zero_()
        self.step.zero_()
        c = self.M // 2
        self.W[:, 0, 0, c] = 1.0
        self.W[:, 1, 1, c] = 1.0
        self.h[:, 0, 0, 0, c] = 1.0
        self.h[:, 1, 1, 0, c] = 1.0

    def set_state(self, W=None, h=None):
        if W is not None:
            self.W.copy_(_f32(W, self.device).expand_as(self.W))
        if h is not None:
            self.h.copy_(_f32(h, self.device).expand_as(self.h))

    _STATE = ("W", "h", "mW", "vW", "mh", "vh", "step")

    def state_dict(self):
        """Everything a run carries from one minibatch to the next (taps, channel estimate, both Adam groups' moments, step counts), on the
        CPU: a sweep can be checkpointed between frames and resumed bit-identically (the reference persists nothing mid-run)."""
        return {k: getattr(self, k).detach().cpu().clone() for k in self._STATE}

    def load_state_dict(self, sd):
        for k in self._STATE:
            t = getattr(self, k)
            if tuple(sd[k].shape) != tuple(t.shape) or sd[k].dtype != t.dtype:
                raise ValueError(f"state {k!r}: expected {tuple(t.shape)} {t.dtype}, got {tuple(sd[k].shape)} {sd[k].dtype}")
            t.copy_(sd[k].to(self.device))

    def train(self, rx, B, steps, lr_W, lr_h=None, stride=None, keep_off=0, keep_len=None, want_q=True, want_y=True,
              want_loss=True, debug_grads=False, no_update=False, want_compact=False):
        """Run ``steps`` minibatch steps per frame on rx[R, n_frames, 2, 2, S] (or [R, 2, 2, S]).

        VAE-LE: defaults (stride = keep_len = B).  VAEflex: stride = keep_len = flex_step, keep_off = (B-flex_step)//2.
        Returns dict of device tensors: q [R,F,2,2n,steps*keep_len], y [R,F,2,2,...], loss [R,F,steps], var_est [R,F,2,steps].
        want_compact adds eq [R,F,2,No] (E_q[x_I] per polarisation) and dec [R,F,2,2,No] int8 (argmax of q per axis): everything the
        per-frame epilogue reads of q, so a sweep can run with want_q=False (dp_epilogue_compact).
        """
        if rx.dim() == 4:
            rx = rx.unsqueeze(1)
        R, F, S = rx.shape[0], rx.shape[1], rx.shape[-1]
        if R != self.R or tuple(rx.shape[2:4]) != (2, 2):
            raise ValueError(f"rx must be [R={self.R}, F, 2, 2, S], got {tuple(rx.shape)}")
        stride = B if stride is None else stride
        keep_len = B if keep_len is None else keep_len
        lr_h = lr_W if lr_h is None else lr_h
        lrW, lrH = self._per_run(lr_W, ()), self._per_run(lr_h, ())
        No = steps * keep_len
        dev = self.device
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {
            "q": e(R, F, 2, 2 * self.n_lev, No) if want_q else None,
            "y": e(R, F, 2, 2, No) if want_y else None,
            "loss": e(R, F, steps) if want_loss else None,
            "var_est": e(R, F, 2, steps) if want_loss else None,
            "gW": e(R, 2, 4, self.M) if debug_grads else None,
            "gh": e(R, 2, 2, 2, self.M) if debug_grads else None,
            "eq": e(R, F, 2, No) if want_compact else None,
            "dec": torch.empty(R, F, 2, 2, No, dtype=torch.int8, device=dev) if want_compact else None,
        }
        a = nat.DPArgs(R=R, n_frames=F, steps=steps, B=B, sps=self.sps, M=self.M, n_lev=self.n_lev, stride_sym=stride,
                       keep_off=keep_off, keep_len=keep_len, S=S, rx=nat.ptr(rx), W=nat.ptr(self.W), h=nat.ptr(self.h),
                       adam_mW=nat.ptr(self.mW), adam_vW=nat.ptr(self.vW), adam_mh=nat.ptr(self.mh), adam_vh=nat.ptr(self.vh),
                       step=nat.ptr(self.step, torch.int32), amp=nat.ptr(self.amp), P=nat.ptr(self.P), var=nat.ptr(self.var),
                       nu_sc=nat.ptr(self.nu_sc), lr_W=nat.ptr(lrW), lr_h=nat.ptr(lrH), q_out=nat.ptr(out["q"]),
                       y_out=nat.ptr(out["y"]), loss=nat.ptr(out["loss"]), var_est=nat.ptr(out["var_est"]),
                       eq_out=nat.ptr(out["eq"]), dec_out=nat.ptr(out["dec"], torch.int8),
                       dbg_gW=nat.ptr(out["gW"]), dbg_gh=nat.ptr(out["gh"]), threads=self.threads, no_update=int(no_update))
        with torch.cuda.device(dev):
            nat.check(nat.lib().vaeq_dp_train(C.byref(a), nat.current_stream(dev)), "vaeq_dp_train")
        out["_keepalive"] = (lrW, lrH, rx)
        return out


def soft_demap(y, amp_levels, var, nu_sc):
    """soft_dec on device: y[R,2,2,N] (or [2,2,N]) -> q[R,2,2n,N]."""
    squeeze = y.dim() == 3
    if squeeze:
        y = y.unsqueeze(0)
    dev, R, N = y.device, y.shape[0], y.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1)
    n = amp.numel()
    var = _f32(var, dev).expand(R, 2).contiguous()
    nu = _f32(nu_sc, dev).expand(R).contiguous()
    y = y.contiguous()
    q = torch.empty(R, 2, 2 * n, N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_soft_demap(R, N, n, nat.ptr(y), nat.ptr(amp), nat.ptr(var), nat.ptr(nu), nat.ptr(q),
                                            nat.current_stream(dev)), "vaeq_soft_demap")
    return q[0] if squeeze else q


def dp_forward(x, W, amp_levels, var, nu_sc, sps=2, want_q=True):
    """twoXtwoFIR.forward (eval) on device: x[R,2,2,N*sps], W[R,2,4,M] -> (q[R,2,2n,N] or None, y[R,2,2,N])."""
    squeeze = x.dim() == 3
    if squeeze:
        x, W = x.unsqueeze(0), W.unsqueeze(0)
    dev, R = x.device, x.shape[0]
    N, M = x.shape[-1] // sps, W.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1)
    n = amp.numel()
    var = _f32(var, dev).expand(R, 2).contiguous()
    nu = _f32(nu_sc, dev).expand(R).contiguous()
    x, W = x.contiguous(), W.contiguous()
    q = torch.empty(R, 2, 2 * n, N, dtype=torch.float32, device=dev) if want_q else None
    y = torch.empty(R, 2, 2, N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_dp_forward(R, N, sps, M, n, nat.ptr(x), nat.ptr(W), nat.ptr(amp), nat.ptr(var), nat.ptr(nu),
                                            nat.ptr(q), nat.ptr(y), nat.current_stream(dev)), "vaeq_dp_forward")
    if squeeze:
        return (q[0] if want_q else None), y[0]
    return q, y


def dp_loss(q, rx, h, amp_levels, P):
    """loss_function_shaping values on device: q[R,2,2n,B], rx[R,2,2,B*sps], h[R,2,2,2,M] (or unbatched) -> (loss, var_est)."""
    squeeze = q.dim() == 3
    if squeeze:
        q, rx, h = q.unsqueeze(0), rx.unsqueeze(0), h.unsqueeze(0)
    dev, R, B = q.device, q.shape[0], q.shape[-1]
    sps, M = rx.shape[-1] // B, h.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1)
    n = amp.numel()
    Pt = _f32(P, dev)
    Pt = (Pt.expand(R, n) if Pt.dim() == 1 else Pt).contiguous()
    q, rx, h = q.contiguous(), rx.contiguous(), h.contiguous()
    loss = torch.empty(R, dtype=torch.float32, device=dev)
    ve = torch.empty(R, 2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_dp_loss(R, B, sps, M, n, nat.ptr(q), nat.ptr(rx), nat.ptr(h), nat.ptr(amp), nat.ptr(Pt),
                                         nat.ptr(loss), nat.ptr(ve), nat.current_stream(dev)), "vaeq_dp_loss")
    return (loss[0], ve[0]) if squeeze else (loss, ve)


class AWGNEngine:
    """R single-polarisation VAE-LE runs (AWGN_channel/func_VAELE_MQAM_shaping.py:275-286): twoFIR weight, h_est and the
    two Adam(amsgrad=True) groups, device-resident."""

    def __init__(self, R, M_est, amp_levels, P, amp_mean, var, device="cuda:0", sps=2, threads=0):
        if M_est % 2 == 0:
            raise ValueError("M_est must be odd")
        self.device = torch.device(device)
        self.R, self.M, self.sps, self.threads = int(R), int(M_est), int(sps), int(threads)
        self.amp = _f32(amp_levels, self.device).reshape(-1)
        self.n_lev = self.amp.numel()
        pr = lambda x, shape: DPEngine._per_run(self, x, shape)
        self.P, self.amp_mean, self.var = pr(P, (self.n_lev,)), pr(amp_mean, ()), pr(var, ())
        z = lambda: torch.zeros(R, 2, self.M, dtype=torch.float32, device=self.device)
        self.W, self.h = z(), z()
        self.mW, self.vW, self.xW, self.mh, self.vh, self.xh = z(), z(), z(), z(), z(), z()
        self.step = torch.zeros(R, dtype=torch.int32, device=self.device)
        self.W[:, 0, self.M // 2] = 1.0   # nn.init.dirac_ (:210)
        self.h[:, 0, self.M // 2] = 1.0   # (:279)

    def set_state(self, W=None, h=None):
        if W is not None:
            self.W.copy_(_f32(W, self.device).reshape(-1, 2, self.M).expand_as(self.W))
        if h is not None:
            self.h.copy_(_f32(h, self.device).expand_as(self.h))

    def train(self, rx, B, steps, lr, want_q=False, want_y=False, debug_grads=False, no_update=False):
        """rx[R,2,S] -> dict(loss[R,steps], q[R,2n,steps*B]?, y[R,2,steps*B]?, gW?, gh?)."""
        R, S = rx.shape[0], rx.shape[-1]
        if R != self.R or rx.shape[1] != 2:
            raise ValueError(f"rx must be [R={self.R}, 2, S], got {tuple(rx.shape)}")
        lr_t = DPEngine._per_run(self, lr, ())
        dev = self.device
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {"loss": e(R, steps), "q": e(R, 2 * self.n_lev, steps * B) if want_q else None,
               "y": e(R, 2, steps * B) if want_y else None, "gW": e(R, 2, self.M) if debug_grads else None,
               "gh": e(R, 2, self.M) if debug_grads else None}
        a = nat.AWGNArgs(R=R, steps=steps, B=B, sps=self.sps, M=self.M, n_lev=self.n_lev, S=S, rx=nat.ptr(rx), W=nat.ptr(self.W),
                         h=nat.ptr(self.h), adam_mW=nat.ptr(self.mW), adam_vW=nat.ptr(self.vW), adam_xW=nat.ptr(self.xW),
                         adam_mh=nat.ptr(self.mh), adam_vh=nat.ptr(self.vh), adam_xh=nat.ptr(self.xh),
                         step=nat.ptr(self.step, torch.int32), amp=nat.ptr(self.amp), P=nat.ptr(self.P),
                         amp_mean=nat.ptr(self.amp_mean), var=nat.ptr(self.var), lr=nat.ptr(lr_t), q_out=nat.ptr(out["q"]),
                         y_out=nat.ptr(out["y"]), loss=nat.ptr(out["loss"]), dbg_gW=nat.ptr(out["gW"]), dbg_gh=nat.ptr(out["gh"]),
                         threads=self.threads, no_update=int(no_update))
        with torch.cuda.device(dev):
            nat.check(nat.lib().vaeq_awgn_train(C.byref(a), nat.current_stream(dev)), "vaeq_awgn_train")
        out["_keepalive"] = (lr_t, rx)
        return out

    def forward(self, x, want_q=True):
        """Validation pass (:313): x[R,2,N*sps] -> (q[R,2n,N] or None, y[R,2,N])."""
        R, N = x.shape[0], x.shape[-1] // self.sps
        x = x.contiguous()
        q = torch.empty(R, 2 * self.n_lev, N, dtype=torch.float32, device=self.device) if want_q else None
        y = torch.empty(R, 2, N, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().vaeq_awgn_forward(R, N, self.sps, self.M, self.n_lev, nat.ptr(x), nat.ptr(self.W), nat.ptr(self.amp),
                                                  nat.ptr(self.amp_mean), nat.ptr(self.var), nat.ptr(q), nat.ptr(y),
                                                  nat.current_stream(self.device)), "vaeq_awgn_forward")
        return q, y

    def validate(self, x, data, n_shift=21):
        """Fused validation pass (:308-318) -> (SER[R] f32, shift[R] i32, y[R,2,N]): forward, find_shift and SER_q in one
        kernel (vaeq_awgn_validate); x[R,2,N*sps] f32, data[R,2,N] f16."""
        R, N = x.shape[0], x.shape[-1] // self.sps
        if tuple(data.shape) != (R, 2, N):
            raise ValueError(f"data must be [R={R}, 2, N={N}], got {tuple(data.shape)}")
        x, data = x.contiguous(), data.contiguous()
        ser = torch.empty(R, dtype=torch.float32, device=self.device)
        shift = torch.empty(R, dtype=torch.int32, device=self.device)
        y = torch.empty(R, 2, N, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().vaeq_awgn_validate(R, N, self.sps, self.M, self.n_lev, int(n_shift), nat.ptr(x), nat.ptr(self.W),
                                                   nat.ptr(self.amp), nat.ptr(self.amp_mean), nat.ptr(self.var),
                                                   nat.ptr(data, torch.float16), nat.ptr(y), nat.ptr(ser), nat.ptr(shift, torch.int32),
                                                   nat.current_stream(self.device)), "vaeq_awgn_validate")
        return ser, shift, y

    def validate_clean(self, frame, n_shift=21, return_sigma=False):
        """validate() on a channel.CleanAwgnFrame: the noise of the frame's (seed, frame index) is added while the samples are staged
        (vaeq_awgn_validate_gen) -- bit for bit validate(*generate_awgn_batch_hip(...)) without the noisy frame's round trip through HBM."""
        import ctypes as C
        R, N = frame.R, frame.N
        if R != self.R or frame.sps != self.sps:
            raise ValueError(f"frame of {R} runs at {frame.sps} sps for an engine of {self.R} runs at {self.sps} sps")
        ser = torch.empty(R, dtype=torch.float32, device=self.device)
        shift = torch.empty(R, dtype=torch.int32, device=self.device)
        y = torch.empty(R, 2, N, dtype=torch.float32, device=self.device)
        sigma = torch.empty(R, dtype=torch.float32, device=self.device) if return_sigma else None
        with torch.cuda.device(self.device):
            nat.check(nat.lib().vaeq_awgn_validate_gen(R, N, self.sps, self.M, self.n_lev, int(n_shift), nat.ptr(frame.sig), frame.Ls,
                                                       nat.ptr(frame.power), nat.ptr(frame.snr),
                                                       None if frame.sigma_fixed is None else nat.ptr(frame.sigma_fixed),
                                                       C.c_uint64(frame.seed), C.c_uint32(frame.frame), nat.ptr(self.W), nat.ptr(self.amp),
                                                       nat.ptr(self.amp_mean), nat.ptr(self.var), nat.ptr(frame.data, torch.float16), nat.ptr(y),
                                                       nat.ptr(ser), nat.ptr(shift, torch.int32), None if sigma is None else nat.ptr(sigma),
                                                       nat.current_stream(self.device)), "vaeq_awgn_validate_gen")
        return (ser, shift, y, sigma) if return_sigma else (ser, shift, y)


class NNEngine:
    """R independent AWGN VAE-NN runs (AWGN_channel/func_VAENN_MQAM.py): flat per-run parameter vectors
    theta = [fc1.weight | fc1.bias | fc2.weight | fc2.bias | h_est] and their AMSGrad state on the device."""

    def __init__(self, R, M_est, kernel_1, kernel_2, amp_levels, device="cuda:0", sps=2, batch_norm=False):
        self.device = torch.device(device)
        self.amp = _f32(amp_levels, self.device).reshape(-1).contiguous()
        self.R, self.M, self.k1, self.k2, self.sps, self.n_lev = int(R), int(M_est), int(kernel_1), int(kernel_2), int(sps), self.amp.numel()
        self.batch_norm = bool(batch_norm)                      # Net_BN (:190-211): theta gains [gamma | beta], bn = running statistics
        C_ = 2 * self.n_lev
        self.bn = torch.cat([torch.zeros(self.R, C_), torch.ones(self.R, C_)], 1).to(self.device).contiguous() if self.batch_norm else None
        NP = int(nat.lib().vaeq_nn_param_count(self.M, self.n_lev, self.k1, self.k2, int(self.batch_norm)))
        if NP < 0:
            raise ValueError(f"unsupported VAE-NN shape M={M_est} k1={kernel_1} k2={kernel_2} n_lev={self.n_lev}: "
                             + nat.lib().vaeq_strerror(NP).decode())
        self.NP = NP
        z = lambda: torch.zeros(self.R, NP, dtype=torch.float32, device=self.device)
        self.theta, self.m, self.v, self.vmax = z(), z(), z(), z()
        self.step = torch.zeros(self.R, dtype=torch.int32, device=self.device)

    def offsets(self):
        C_ = 2 * self.n_lev
        o = [0, C_ * 2 * self.k1]
        o += [o[-1] + C_, o[-1] + C_ + C_ * C_ * self.k2]
        o += [o[-1] + C_]
        if self.batch_norm:
            o += [o[-1] + C_, o[-1] + 2 * C_]
        o += [o[-1] + 2 * self.M]
        return o

    def init_parameters(self, generator=None):
        """nn.init.xavier_uniform_ on both conv weights, PyTorch's default uniform(+-1/sqrt(fan_in)) on the biases (:172-176), Dirac
        h_est (:244-246), independently per run."""
        C_, o = 2 * self.n_lev, self.offsets()
        u = lambda n, bound: (torch.rand(self.R, n, generator=generator, device=self.device) * 2 - 1) * bound
        # Net: xavier_uniform_ on fc1 (fan_in 2 k1, fan_out C k1, :173); Net_BN: kaiming_uniform_ (a = 0: bound sqrt(6 / fan_in), :195)
        self.theta[:, o[0]:o[1]] = u(o[1] - o[0], (6.0 / (2 * self.k1)) ** 0.5 if self.batch_norm else (6.0 / (2 * self.k1 + C_ * self.k1)) ** 0.5)
        self.theta[:, o[1]:o[2]] = u(C_, (2 * self.k1) ** -0.5)
        self.theta[:, o[2]:o[3]] = u(o[3] - o[2], (6.0 / (2 * C_ * self.k2)) ** 0.5)
        self.theta[:, o[3]:o[4]] = u(C_, (C_ * self.k2) ** -0.5)
        if self.batch_norm:
            self.theta[:, o[4]:o[5]] = 1                        # BatchNorm weight
            self.theta[:, o[5]:o[6]] = 0                        # BatchNorm bias
            self.bn[:, :C_] = 0
            self.bn[:, C_:] = 1
        oh = o[-2]
        self.theta[:, oh:] = 0
        self.theta[:, oh + self.M // 2] = 1
        for t in (self.m, self.v, self.vmax):
            t.zero_()
        self.step.zero_()

    def train(self, rx, B, steps, lr, want_q=False, debug_grads=False, no_update=False):
        """rx[R,2,S] -> dict(loss[R,steps], q[R,2n,steps*B]?, g[R,NP]?)."""
        R, S = rx.shape[0], rx.shape[-1]
        if R != self.R or rx.dim() != 3 or rx.shape[1] != 2:
            raise ValueError(f"rx must be [R={self.R}, 2, S], got {tuple(rx.shape)}")
        rx = rx.contiguous()
        dev = self.device
        lr_t = _f32(lr, dev).expand(R).contiguous()
        out = {"loss": torch.empty(R, steps, dtype=torch.float32, device=dev),
               "q": torch.empty(R, 2 * self.n_lev, steps * B, dtype=torch.float32, device=dev) if want_q else None,
               "g": torch.empty(R, self.NP, dtype=torch.float32, device=dev) if debug_grads else None}
        a = nat.NNArgs(R=R, steps=steps, B=B, sps=self.sps, M=self.M, n_lev=self.n_lev, k1=self.k1, k2=self.k2, S=S, rx=nat.ptr(rx),
                       theta=nat.ptr(self.theta), adam_m=nat.ptr(self.m), adam_v=nat.ptr(self.v), adam_x=nat.ptr(self.vmax),
                       step=nat.ptr(self.step, torch.int32), amp=nat.ptr(self.amp), lr=nat.ptr(lr_t), loss=nat.ptr(out["loss"]),
                       q_out=nat.ptr(out["q"]), dbg_g=nat.ptr(out["g"]), no_update=int(no_update), batch_norm=int(self.batch_norm),
                       bn_running=nat.ptr(self.bn))
        with torch.cuda.device(dev):
            nat.check(nat.lib().vaeq_nn_train(C.byref(a), nat.current_stream(dev)), "vaeq_nn_train")
        out["_keepalive"] = (lr_t, rx)
        return out

    def forward(self, x):
        """Validation pass (:293-295): x[R,2,N*sps] -> q[R,2n,N]."""
        R, N = x.shape[0], x.shape[-1] // self.sps
        x = x.contiguous()
        q = torch.empty(R, 2 * self.n_lev, N, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().vaeq_nn_forward(R, N, self.sps, self.M, self.n_lev, self.k1, self.k2, nat.ptr(x), nat.ptr(self.theta),
                                                nat.ptr(self.bn), nat.ptr(q), nat.current_stream(self.device)), "vaeq_nn_forward")
        return q

    def validate(self, x, data, n_shift=21):
        """Fused validation pass (:287-301) -> (SER[R] f32, shift[R] i32): eval forward, find_shift and SER_q in one kernel
        (vaeq_nn_validate); x[R,2,N*sps] f32, data[R,2,N] f16."""
        R, N = x.shape[0], x.shape[-1] // self.sps
        if tuple(data.shape) != (R, 2, N):
            raise ValueError(f"data must be [R={R}, 2, N={N}], got {tuple(data.shape)}")
        x, data = x.contiguous(), data.contiguous()
        ser = torch.empty(R, dtype=torch.float32, device=self.device)
        shift = torch.empty(R, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().vaeq_nn_validate(R, N, self.sps, self.M, self.n_lev, self.k1, self.k2, int(n_shift), nat.ptr(x),
                                                 nat.ptr(self.theta), nat.ptr(self.bn), nat.ptr(self.amp), nat.ptr(data, torch.float16), nat.ptr(ser),
                                                 nat.ptr(shift, torch.int32), nat.current_stream(self.device)), "vaeq_nn_validate")
        return ser, shift


def dp_epilogue(q, y, data, amp_levels, nu_sc, var, batch_len=None):
    """Per-frame epilogue on the device (vaeq_dp_epilogue): q[R,2,2n,N], y[R,2,2,N], data[R,2,2,N] fp16 ->
    dict(SER[R,4], shift_q[R,2], r_q[R], shift_c[R,2], r_c[R]).  batch_len None = VAEflex (no per-minibatch cut)."""
    dev, R, N = q.device, q.shape[0], q.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1)
    n = amp.numel()
    var = _f32(var, dev).expand(R, 2).contiguous()
    nu = _f32(nu_sc, dev).expand(R).contiguous()
    q, y = q.contiguous(), y.contiguous()
    data = data.to(torch.float16).contiguous()
    ser = torch.empty(R, 4, dtype=torch.float32, device=dev)
    shift = torch.empty(R, 2, 2, dtype=torch.int32, device=dev)
    rflag = torch.empty(R, 2, dtype=torch.int32, device=dev)
    ws = torch.empty(int(nat.lib().vaeq_dp_epilogue_ws_bytes(R, N)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_dp_epilogue(R, N, n, int(batch_len or 0), nat.ptr(q), nat.ptr(y), nat.ptr(data, torch.float16),
                                             nat.ptr(amp), nat.ptr(var), nat.ptr(nu), nat.ptr(ser), nat.ptr(shift, torch.int32),
                                             nat.ptr(rflag, torch.int32), nat.ptr(ws, torch.uint8), nat.current_stream(dev)),
                  "vaeq_dp_epilogue")
    return dict(SER=ser, shift_q=shift[:, 0].long(), r_q=rflag[:, 0].long(), shift_c=shift[:, 1].long(), r_c=rflag[:, 1].long())


def dp_epilogue_compact(eq, dec, y, data, amp_levels, nu_sc, var, batch_len=None):
    """dp_epilogue fed by the training kernel's compact outputs of one frame (vaeq_dp_epilogue_compact): eq[R,2,N] f32,
    dec[R,2,2,N] int8, y[R,2,2,N], data[R,2,2,N] fp16 -> the same dict; bit-identical to dp_epilogue on that call's q."""
    dev, R, N = y.device, y.shape[0], y.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1)
    var = _f32(var, dev).expand(R, 2).contiguous()
    nu = _f32(nu_sc, dev).expand(R).contiguous()
    eq, dec, y = eq.contiguous(), dec.contiguous(), y.contiguous()
    data = data.to(torch.float16).contiguous()
    ser = torch.empty(R, 4, dtype=torch.float32, device=dev)
    shift = torch.empty(R, 2, 2, dtype=torch.int32, device=dev)
    rflag = torch.empty(R, 2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_dp_epilogue_compact(R, N, amp.numel(), int(batch_len or 0), nat.ptr(eq), nat.ptr(dec, torch.int8), nat.ptr(y),
                                                     nat.ptr(data, torch.float16), nat.ptr(amp), nat.ptr(var), nat.ptr(nu), nat.ptr(ser),
                                                     nat.ptr(shift, torch.int32), nat.ptr(rflag, torch.int32), nat.current_stream(dev)),
                  "vaeq_dp_epilogue_compact")
    return dict(SER=ser, shift_q=shift[:, 0].long(), r_q=rflag[:, 0].long(), shift_c=shift[:, 1].long(), r_c=rflag[:, 1].long())


def cma_epilogue(y, data, amp_levels, nu_sc, var):
    """The constant-modulus baselines' two-stage epilogue of one frame in one launch (vaeq_cma_epilogue): y[R,2,2,N] = phase-corrected output cut
    to [10:-10], data[R,2,2,N] fp16 cut likewise -> dict(SER[R,4] (constellation rows, then soft-demapper rows), shift_c, r_c, shift_q, r_q)."""
    dev, R, N = y.device, y.shape[0], y.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1)
    var = _f32(var, dev).expand(R, 2).contiguous()
    nu = _f32(nu_sc, dev).expand(R).contiguous()
    y = y.contiguous()
    data = data.to(torch.float16).contiguous()
    ser = torch.empty(R, 4, dtype=torch.float32, device=dev)
    shift = torch.empty(R, 2, 2, dtype=torch.int32, device=dev)
    rflag = torch.empty(R, 2, dtype=torch.int32, device=dev)
    L = nat.lib()
    ws = torch.empty(int(L.vaeq_dp_epilogue_ws_bytes(R, N)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.vaeq_cma_epilogue(R, N, amp.numel(), nat.ptr(y), nat.ptr(data, torch.float16), nat.ptr(amp), nat.ptr(var), nat.ptr(nu), nat.ptr(ser),
                                      nat.ptr(shift, torch.int32), nat.ptr(rflag, torch.int32), nat.ptr(ws, torch.uint8), nat.current_stream(dev)),
                  "vaeq_cma_epilogue")
    return dict(SER=ser, shift_q=shift[:, 0].long(), r_q=rflag[:, 0].long(), shift_c=shift[:, 1].long(), r_c=rflag[:, 1].long())


def awgn_loss(q, x, h, amp_levels, P=None):
    """ELBO of the single-polarisation variants for a given q (vaeq_awgn_loss): q[R,2n,B] (or [2n,B]), x[R,2,B*sps], h[R,2,M];
    P[R,n] / [n] -> the VAE-LE form (KL to the prior), P None -> the VAE-NN form (entropy).  Returns loss[R] (or a 0-dim tensor)."""
    single = q.dim() == 2
    if single:
        q, x, h = q.unsqueeze(0), x.unsqueeze(0), h.unsqueeze(0)
    dev, R, B = q.device, q.shape[0], q.shape[-1]
    amp = _f32(amp_levels, dev).reshape(-1).contiguous()
    n = amp.numel()
    q, x, h = q.contiguous().float(), x.contiguous().float(), h.contiguous().float()
    Pt = None if P is None else _f32(P, dev).expand(R, n).contiguous()
    loss = torch.empty(R, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_awgn_loss(R, B, x.shape[-1] // B, h.shape[-1], n, nat.ptr(q), nat.ptr(x), nat.ptr(h), nat.ptr(amp),
                                           nat.ptr(Pt), nat.ptr(loss), nat.current_stream(dev)), "vaeq_awgn_loss")
    return loss[0] if single else loss


def cma(rx, h, lr, sps=2, mode="CMA", batch_len=100, symb_step=10, R_mod=1.0, want_e=True):
    """CMA / CMAbatch / CMAflex (shared_funcs.py:341-488, vaeq_cma) for R runs: rx[R,2,2,N], h[R,2,2,2,M] (updated IN PLACE),
    lr scalar or [R] -> (out[R,2,2,N//sps], e[R,N//sps,2] or None)."""
    dev, R, N = rx.device, rx.shape[0], rx.shape[-1]
    if tuple(rx.shape[1:3]) != (2, 2) or tuple(h.shape[:4]) != (R, 2, 2, 2) or not h.is_contiguous():
        raise ValueError(f"rx must be [R,2,2,N] and h a contiguous [R,2,2,2,M], got {tuple(rx.shape)}, {tuple(h.shape)}")
    m = {"CMA": 0, "CMAbatch": 1, "CMAflex": 1}[mode]
    step = batch_len if mode == "CMAbatch" else symb_step
    rx = rx.contiguous()
    lr_t = _f32(lr, dev).expand(R).contiguous()
    K = N // sps
    out = torch.empty(R, 2, 2, K, dtype=torch.float32, device=dev)
    e = torch.empty(R, K, 2, dtype=torch.float32, device=dev) if want_e else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_cma(R, N, sps, h.shape[-1], m, int(batch_len), int(step), nat.ptr(rx), float(R_mod), nat.ptr(h),
                                     nat.ptr(lr_t), nat.ptr(out), nat.ptr(e), nat.current_stream(dev)), "vaeq_cma")
    return out, e


def cpe(y, M_ma=501):
    """Viterbi-Viterbi carrier phase estimation (shared_funcs.py:139-186, vaeq_cpe): y[R,2,2,N] (or [2,2,N]) -> corrected y."""
    squeeze = y.dim() == 3
    if squeeze:
        y = y.unsqueeze(0)
    y = y.contiguous().float()
    out = torch.empty_like(y)
    with torch.cuda.device(y.device):
        nat.check(nat.lib().vaeq_cpe(y.shape[0], y.shape[-1], int(M_ma), nat.ptr(y), nat.ptr(out), nat.current_stream(y.device)), "vaeq_cpe")
    return out[0] if squeeze else out
