"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/libvaeq_oracle.so.

Every wrapper takes/returns numpy arrays; ``dtype`` selects the float (reference
arithmetic) or double ("truth") instantiation of the same C source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/libvaeq_oracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "libvaeq_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("vaeq_oracle.c", "vaeq_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvaeq_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", C.c_float, np.float32
    if dtype == np.float64:
        return "_f64", C.c_double, np.float64
    raise TypeError(dtype)


def _arr(a, npt):
    return np.ascontiguousarray(a, dtype=npt)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def dp_forward(x, W, amp, var, nu_sc, sps=2, dtype=np.float32):
    """twoXtwoFIR.forward (shared_funcs.py:500-527) -> (q[2,2n,B], out[2,2,B])."""
    sfx, ct, npt = _sfx(dtype)
    x, W, amp, var = _arr(x, npt), _arr(W, npt), _arr(amp, npt), _arr(var, npt)
    L, M, n = x.shape[-1], W.shape[-1], amp.shape[0]
    B = L // sps
    q, out = np.empty((2, 2 * n, B), npt), np.empty((2, 2, B), npt)
    f = getattr(lib(), "vaeq_oracle_dp_forward" + sfx)
    f.restype = None
    f(B, sps, M, n, _p(x), _p(W), _p(amp), _p(var), ct(nu_sc), _p(q), _p(out))
    return q, out


def dp_soft_dec(out, var, amp, nu_sc, dtype=np.float32):
    """soft_dec (shared_funcs.py:529-542)."""
    sfx, ct, npt = _sfx(dtype)
    out, amp, var = _arr(out, npt), _arr(amp, npt), _arr(var, npt)
    N, n = out.shape[-1], amp.shape[0]
    q = np.empty((2, 2 * n, N), npt)
    f = getattr(lib(), "vaeq_oracle_dp_soft_dec" + sfx)
    f.restype = None
    f(N, n, _p(out), _p(var), _p(amp), ct(nu_sc), _p(q))
    return q


def dp_loss(q, x, h, amp, P, dtype=np.float32):
    """loss_function_shaping (shared_funcs.py:92-137) -> (loss, var_est[2])."""
    sfx, ct, npt = _sfx(dtype)
    q, x, h, amp, P = _arr(q, npt), _arr(x, npt), _arr(h, npt), _arr(amp, npt), _arr(P, npt)
    B, L, M, n = q.shape[-1], x.shape[-1], h.shape[-1], amp.shape[0]
    ve = np.empty(2, npt)
    f = getattr(lib(), "vaeq_oracle_dp_loss" + sfx)
    f.restype = ct
    loss = f(B, L // B, M, n, _p(q), _p(x), _p(h), _p(amp), _p(P), _p(ve))
    return npt(loss), ve


def dp_step_grads(x, W, h, amp, P, var, nu_sc, sps=2, dtype=np.float32):
    """forward + loss + backward of one minibatch -> dict(q,out,loss,var_est,gW,gh)."""
    sfx, ct, npt = _sfx(dtype)
    x, W, h, amp, P, var = (_arr(a, npt) for a in (x, W, h, amp, P, var))
    L, M, n = x.shape[-1], W.shape[-1], amp.shape[0]
    B = L // sps
    q, out, ve = np.empty((2, 2 * n, B), npt), np.empty((2, 2, B), npt), np.empty(2, npt)
    gW, gh = np.empty((2, 4, M), npt), np.empty((2, 2, 2, M), npt)
    f = getattr(lib(), "vaeq_oracle_dp_step_grads" + sfx)
    f.restype = ct
    loss = f(B, sps, M, n, _p(x), _p(W), _p(h), _p(amp), _p(P), _p(var), ct(nu_sc), _p(q), _p(out), _p(ve), _p(gW), _p(gh))
    return dict(q=q, out=out, loss=npt(loss), var_est=ve, gW=gW, gh=gh)


def adam(p, g, m, v, step, lr, vmax=None, dtype=np.float32):
    """torch.optim.Adam single-tensor update, in place on p, m, v (and vmax for amsgrad)."""
    sfx, ct, npt = _sfx(dtype)
    for a in (p, m, v):
        assert a.dtype == npt and a.flags.c_contiguous
    g = _arr(g, npt)
    f = getattr(lib(), "vaeq_oracle_adam" + sfx)
    f.restype = None
    f(p.size, _p(p), _p(g), _p(m), _p(v), _p(vmax), int(step), C.c_double(lr), int(vmax is not None))


class DPState:
    """Caller-owned optimiser state of one DP run (taps, channel estimate, Adam moments, step count)."""

    def __init__(self, M, dtype=np.float32, W=None, h=None):
        npt = np.dtype(dtype).type
        self.W = np.zeros((2, 4, M), npt)
        self.h = np.zeros((2, 2, 2, M), npt)
        self.W[0, 0, M // 2] = self.W[1, 1, M // 2] = 1       # nn.init.dirac_, shared_funcs.py:495
        self.h[0, 0, 0, M // 2] = self.h[1, 1, 0, M // 2] = 1  # shared_funcs.py:585
        if W is not None:
            self.W[...] = W
        if h is not None:
            self.h[...] = h
        self.mW, self.vW = np.zeros_like(self.W), np.zeros_like(self.W)
        self.mh, self.vh = np.zeros_like(self.h), np.zeros_like(self.h)
        self.step = C.c_int(0)


def dp_train(state, rx, n_steps, B, amp, P, var, nu_sc, lr_W, lr_h, sps=2, stride=None, keep_off=0, keep_len=None,
             want_q=True, dtype=np.float32):
    """One frame of the DP minibatch loop (VAE-LE: stride=B; VAEflex: stride=flex_step, centre slice kept)."""
    sfx, ct, npt = _sfx(dtype)
    rx, amp, P, var = (_arr(a, npt) for a in (rx, amp, P, var))
    stride = B if stride is None else stride
    keep_len = B if keep_len is None else keep_len
    S, M, n = rx.shape[-1], state.W.shape[-1], amp.shape[0]
    assert (n_steps - 1) * stride * sps + B * sps <= S
    No = n_steps * keep_len
    q_out = np.empty((2, 2 * n, No), npt) if want_q else None
    y_out = np.empty((2, 2, No), npt)
    loss, ve = np.empty(n_steps, npt), np.empty((2, n_steps), npt)
    f = getattr(lib(), "vaeq_oracle_dp_train" + sfx)
    f.restype = None
    f(n_steps, B, sps, M, n, stride, keep_off, keep_len, S, _p(rx), _p(state.W), _p(state.h), _p(state.mW), _p(state.vW),
      _p(state.mh), _p(state.vh), C.byref(state.step), _p(amp), _p(P), _p(var), ct(nu_sc), C.c_double(lr_W), C.c_double(lr_h),
      _p(q_out), _p(y_out), _p(loss), _p(ve))
    return dict(q=q_out, out=y_out, loss=loss, var_est=ve)


def dp_train_batch_f32(R, n_threads, n_steps, B, sps, M, n, stride, keep_off, keep_len, rx, W, h, mW, vW, mh, vh, step,
                       amp, P, var, nu_sc, lr_W, lr_h, q_out, y_out, loss, var_est):
    """OpenMP batch of R independent runs (cpu_baseline leg).  All arrays float32 C-contiguous, step int32[R]."""
    f = lib().vaeq_oracle_dp_train_batch_f32
    f.restype = C.c_int
    S = rx.shape[-1]
    return f(R, n_threads, n_steps, B, sps, M, n, stride, keep_off, keep_len, S, _p(rx), _p(W), _p(h), _p(mW), _p(vW),
             _p(mh), _p(vh), _p(step), _p(amp), _p(P), _p(var), _p(nu_sc), _p(lr_W), _p(lr_h), _p(q_out), _p(y_out),
             _p(loss), _p(var_est))


# ------------------------------------------------------------------ AWGN
def awgn_forward(x, W, amp, amp_mean, var, sps=2, dtype=np.float32):
    """twoFIR.forward (func_VAELE_MQAM_shaping.py:214-231) -> (q[2n,B], out[2,B])."""
    sfx, ct, npt = _sfx(dtype)
    x, W, amp = _arr(x, npt), _arr(W, npt), _arr(amp, npt)
    L, M, n = x.shape[-1], W.shape[-1], amp.shape[0]
    B = L // sps
    q, out = np.empty((2 * n, B), npt), np.empty((2, B), npt)
    f = getattr(lib(), "vaeq_oracle_awgn_forward" + sfx)
    f.restype = None
    f(B, sps, M, n, _p(x), _p(W), _p(amp), ct(amp_mean), ct(var), _p(q), _p(out), None, None)
    return q, out


def awgn_loss(q, x, h, amp, P, dtype=np.float32):
    """loss_function (func_VAELE_MQAM_shaping.py:63-95)."""
    sfx, ct, npt = _sfx(dtype)
    q, x, h, amp, P = (_arr(a, npt) for a in (q, x, h, amp, P))
    B, L, M, n = q.shape[-1], x.shape[-1], h.shape[-1], amp.shape[0]
    f = getattr(lib(), "vaeq_oracle_awgn_loss" + sfx)
    f.restype = ct
    return npt(f(B, L // B, M, n, _p(q), _p(x), _p(h), _p(amp), _p(P)))


def awgn_step_grads(x, W, h, amp, P, amp_mean, var, sps=2, dtype=np.float32):
    sfx, ct, npt = _sfx(dtype)
    x, W, h, amp, P = (_arr(a, npt) for a in (x, W, h, amp, P))
    L, M, n = x.shape[-1], W.shape[-1], amp.shape[0]
    B = L // sps
    q, out = np.empty((2 * n, B), npt), np.empty((2, B), npt)
    gW, gh = np.empty((1, 2, M), npt), np.empty((2, M), npt)
    f = getattr(lib(), "vaeq_oracle_awgn_step_grads" + sfx)
    f.restype = ct
    loss = f(B, sps, M, n, _p(x), _p(W), _p(h), _p(amp), _p(P), ct(amp_mean), ct(var), _p(q), _p(out), _p(gW), _p(gh))
    return dict(q=q, out=out, loss=npt(loss), gW=gW, gh=gh)


class AWGNState:
    def __init__(self, M, dtype=np.float32, W=None, h=None):
        npt = np.dtype(dtype).type
        self.W = np.zeros((1, 2, M), npt)
        self.h = np.zeros((2, M), npt)
        self.W[0, 0, M // 2] = 1   # nn.init.dirac_, func_VAELE_MQAM_shaping.py:210
        self.h[0, M // 2] = 1      # func_VAELE_MQAM_shaping.py:279
        if W is not None:
            self.W[...] = W
        if h is not None:
            self.h[...] = h
        self.mW, self.vW, self.vmaxW = (np.zeros_like(self.W) for _ in range(3))
        self.mh, self.vh, self.vmaxh = (np.zeros_like(self.h) for _ in range(3))
        self.step = C.c_int(0)


def awgn_train(state, rx, n_steps, B, amp, P, amp_mean, var, lr, sps=2, dtype=np.float32):
    sfx, ct, npt = _sfx(dtype)
    rx, amp, P = (_arr(a, npt) for a in (rx, amp, P))
    S, M, n = rx.shape[-1], state.W.shape[-1], amp.shape[0]
    assert n_steps * B * sps <= S
    loss = np.empty(n_steps, npt)
    f = getattr(lib(), "vaeq_oracle_awgn_train" + sfx)
    f.restype = None
    f(n_steps, B, sps, M, n, S, _p(rx), _p(state.W), _p(state.h), _p(state.mW), _p(state.vW), _p(state.vmaxW),
      _p(state.mh), _p(state.vh), _p(state.vmaxh), C.byref(state.step), _p(amp), _p(P), ct(amp_mean), ct(var),
      C.c_double(lr), _p(loss))
    return loss


# ------------------------------------------------------------------ AWGN VAE-NN (row f3)
def nn_param_count(n, k1, k2, M):
    C_ = 2 * n
    return C_ * 2 * k1 + C_ + C_ * C_ * k2 + C_ + 2 * M


def nn_pack(w1, b1, w2, b2, h, dtype=np.float32):
    """Flat parameter vector [fc1.weight | fc1.bias | fc2.weight | fc2.bias | h_est] (func_VAENN_MQAM.py:170-176, 244-246)."""
    return np.concatenate([np.asarray(a, dtype=dtype).reshape(-1) for a in (w1, b1, w2, b2, h)])


def nn_unpack(theta, n, k1, k2, M):
    C_ = 2 * n
    o = np.cumsum([0, C_ * 2 * k1, C_, C_ * C_ * k2, C_, 2 * M])
    return (theta[o[0]:o[1]].reshape(C_, 2, k1), theta[o[1]:o[2]], theta[o[2]:o[3]].reshape(C_, C_, k2), theta[o[3]:o[4]],
            theta[o[4]:o[5]].reshape(2, M))


def nn_forward(x, theta, n, k1, k2, sps=2, dtype=np.float32):
    """Net.forward (func_VAENN_MQAM.py:178-188) -> q[2n, B]."""
    sfx, ct, npt = _sfx(dtype)
    x, theta = _arr(x, npt), _arr(theta, npt)
    L = x.shape[-1]
    B = L // sps
    z1, a2, q = np.empty((2 * n, L), npt), np.empty((2 * n, B), npt), np.empty((2 * n, B), npt)
    f = getattr(lib(), "vaeq_oracle_nn_forward" + sfx)
    f.restype = None
    f(B, sps, n, k1, k2, _p(x), _p(theta), _p(z1), _p(a2), _p(q))
    return q


def nn_loss(q, x, h, amp, dtype=np.float32):
    """loss_function (func_VAENN_MQAM.py:63-95)."""
    sfx, ct, npt = _sfx(dtype)
    q, x, h, amp = (_arr(a, npt) for a in (q, x, h, amp))
    B, L, M, n = q.shape[-1], x.shape[-1], h.shape[-1], amp.shape[0]
    f = getattr(lib(), "vaeq_oracle_nn_loss" + sfx)
    f.restype = ct
    return npt(f(B, L // B, M, n, _p(q), _p(x), _p(h), _p(amp)))


def nn_step_grads(x, theta, amp, k1, k2, M, sps=2, dtype=np.float32):
    sfx, ct, npt = _sfx(dtype)
    x, theta, amp = _arr(x, npt), _arr(theta, npt), _arr(amp, npt)
    L, n = x.shape[-1], amp.shape[0]
    B = L // sps
    assert theta.size == nn_param_count(n, k1, k2, M)
    q, g = np.empty((2 * n, B), npt), np.empty(theta.size, npt)
    f = getattr(lib(), "vaeq_oracle_nn_step_grads" + sfx)
    f.restype = ct
    loss = f(B, sps, M, n, k1, k2, _p(x), _p(theta), _p(amp), _p(q), _p(g))
    return dict(q=q, loss=npt(loss), g=g)


class NNState:
    """Caller-owned state of one VAE-NN run: flat parameters + AMSGrad moments + step count."""

    def __init__(self, theta, dtype=np.float32):
        self.theta = np.array(theta, dtype=dtype).reshape(-1)
        self.m, self.v, self.vmax = (np.zeros_like(self.theta) for _ in range(3))
        self.step = C.c_int(0)


def nn_train(state, rx, n_steps, B, amp, k1, k2, M, lr, sps=2, dtype=np.float32):
    sfx, ct, npt = _sfx(dtype)
    rx, amp = _arr(rx, npt), _arr(amp, npt)
    S, n = rx.shape[-1], amp.shape[0]
    assert n_steps * B * sps <= S and state.theta.size == nn_param_count(n, k1, k2, M)
    loss = np.empty(n_steps, npt)
    f = getattr(lib(), "vaeq_oracle_nn_train" + sfx)
    f.restype = None
    f(n_steps, B, sps, M, n, k1, k2, S, _p(rx), _p(state.theta), _p(state.m), _p(state.v), _p(state.vmax), C.byref(state.step), _p(amp),
      C.c_double(lr), _p(loss))
    return loss


# ---- Net_BN (BatchNorm variant): theta = [w1 | b1 | w2 | b2 | gamma | beta | h], bn = [running_mean | running_var]
def nnbn_param_count(n, k1, k2, M):
    return nn_param_count(n, k1, k2, M) + 4 * n


def nnbn_step_grads(x, theta, bn, amp, k1, k2, M, sps=2, dtype=np.float32):
    """Training-mode forward + loss + backward of Net_BN; returns the UPDATED running statistics too."""
    sfx, ct, npt = _sfx(dtype)
    x, theta, amp = _arr(x, npt), _arr(theta, npt), _arr(amp, npt)
    bn = np.array(bn, dtype=npt).reshape(-1)
    L, n = x.shape[-1], amp.shape[0]
    B = L // sps
    assert theta.size == nnbn_param_count(n, k1, k2, M) and bn.size == 4 * n
    q, g = np.empty((2 * n, B), npt), np.empty(theta.size, npt)
    f = getattr(lib(), "vaeq_oracle_nnbn_step_grads" + sfx)
    f.restype = ct
    loss = f(B, sps, M, n, k1, k2, _p(x), _p(theta), _p(bn), _p(amp), _p(q), _p(g))
    return dict(q=q, loss=npt(loss), g=g, bn=bn)


def nnbn_forward_eval(x, theta, bn, n, k1, k2, sps=2, dtype=np.float32):
    sfx, ct, npt = _sfx(dtype)
    x, theta, bn = _arr(x, npt), _arr(theta, npt), _arr(bn, npt)
    B = x.shape[-1] // sps
    q = np.empty((2 * n, B), npt)
    f = getattr(lib(), "vaeq_oracle_nnbn_forward_eval" + sfx)
    f.restype = None
    f(B, sps, n, k1, k2, _p(x), _p(theta), _p(bn), _p(q))
    return q


class NNBNState(NNState):
    def __init__(self, theta, n, dtype=np.float32):
        super().__init__(theta, dtype)
        self.bn = np.concatenate([np.zeros(2 * n, dtype), np.ones(2 * n, dtype)])     # running_mean = 0, running_var = 1


def nnbn_train(state, rx, n_steps, B, amp, k1, k2, M, lr, sps=2, dtype=np.float32):
    sfx, ct, npt = _sfx(dtype)
    rx, amp = _arr(rx, npt), _arr(amp, npt)
    S, n = rx.shape[-1], amp.shape[0]
    assert n_steps * B * sps <= S and state.theta.size == nnbn_param_count(n, k1, k2, M)
    loss = np.empty(n_steps, npt)
    f = getattr(lib(), "vaeq_oracle_nnbn_train" + sfx)
    f.restype = None
    f(n_steps, B, sps, M, n, k1, k2, S, _p(rx), _p(state.theta), _p(state.bn), _p(state.m), _p(state.v), _p(state.vmax), C.byref(state.step),
      _p(amp), C.c_double(lr), _p(loss))
    return loss


# ------------------------------------------------------------------ row f4: constant-modulus baselines
def cma(rx, h, lr, sps=2, mode="CMA", batchlen=100, symb_step=10, R=1.0, dtype=np.float32):
    """CMA / CMAbatch / CMAflex (shared_funcs.py:341-488) on one frame rx[2,2,N]; h[2,2,2,M] is updated IN PLACE (pass a copy).
    Returns (out[2,2,N//sps], e[N//sps,2])."""
    sfx, ct, npt = _sfx(dtype)
    rx = _arr(rx, npt)
    assert h.dtype == npt and h.flags["C_CONTIGUOUS"]
    N, M = rx.shape[-1], h.shape[-1]
    K = N // sps
    out, e = np.empty((2, 2, K), npt), np.empty((K, 2), npt)
    m = {"CMA": 0, "CMAbatch": 1, "CMAflex": 1}[mode]
    step = batchlen if mode == "CMAbatch" else symb_step
    f = getattr(lib(), "vaeq_oracle_cma" + sfx)
    f.restype = None
    f(N, sps, M, m, batchlen, step, _p(rx), ct(R), _p(h), C.c_double(lr), _p(out), _p(e))
    return out, e
