"""Input producer for the hot path: seeded dual-polarisation / AWGN channel simulators.

``generate_data_shaping`` / ``generate_data`` restate optical_DP_channel/shared_funcs.py:17-90 and
AWGN_channel/func_VAELE_MQAM_shaping.py:28-61 in numpy with the SAME random-number consumption
(``Generator.choice`` for the PCS symbols, legacy global ``randn`` for the noise), so that a run seeded
like tools/capture_golden.py reproduces the reference's data stream (pinned by tests/golden/G6_generator.npz).
The reference seeds nothing; here every call may be given explicit generators.

``generate_batch_gpu`` is the same physical model evaluated with torch ops on the device for a whole batch of
runs at once: it feeds bench.py and large sweeps without a host round trip.  Its random stream
is torch's, not numpy's.
"""
import math

import numpy as np
import torch

PULSE_SPAN = 8      # T: pulse length in symbols        (shared_funcs.py:66)
ROLL_OFF = 0.1      # beta                               (shared_funcs.py:67)


def rrcfir(T, sps, beta):
    """Root-raised-cosine taps, unit energy (shared_funcs.py:27-36).  float32 time grid like the reference."""
    t = np.arange(-T * sps / 2, T * sps / 2, 1 / sps, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        num = np.sin(np.pi * t * (1 - beta)) + 4 * beta * t * np.cos(np.pi * t * (1 + beta))
        h = num / (np.pi * t * (1 - (4 * beta * t) ** 2))
    h[np.abs(t) == 1 / 4 / beta] = beta / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / 4 / beta) + (1 - 2 / np.pi) * np.cos(np.pi / 4 / beta))
    h[t == 0] = 1 + beta * (4 / np.pi - 1)
    return h / np.linalg.norm(h)


def rcfir(T, sps, beta):
    """Raised-cosine taps, unit energy (shared_funcs.py:17-25)."""
    t = np.arange(-T * sps / 2, T * sps / 2, 1 / sps, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.sinc(t) * np.cos(np.pi * beta * t) / (1 - (2 * beta * t) ** 2)
    h[np.abs(t) == 1 / 2 / beta] = np.pi / 4 * np.sinc(1 / (2 * beta))
    return h / np.linalg.norm(h)


def _fiber_matrix(freq, tau_pmd, phiIQ, theta):
    """H(f) = R^T diag(e^{j pi tau f}, e^{-j pi tau f}) R with the IQ-phase folded into R (shared_funcs.py:42-50)."""
    d = np.exp(1j * np.pi * tau_pmd * freq)
    c, s = np.cos(theta), np.sin(theta)
    e = np.exp(-1j * phiIQ)
    R = ((c * e[0], s * e[0]), (-s * e[1], c * e[1]))
    RT = ((c * e[0], -s * e[0]), (s * e[1], c * e[1]))
    di = 1 / d
    return [[RT[a][0] * d * R[0][b] + RT[a][1] * di * R[1][b] for b in range(2)] for a in range(2)]


def simulate_dispersion(rx, symb_rate, sps, tau_cd, tau_pmd, phiIQ, theta):
    """Residual CD + PMD + polarisation rotation + IQ phase in the frequency domain (shared_funcs.py:38-54)."""
    spec = np.fft.fft(rx, axis=1)
    freq = np.fft.fftfreq(rx.shape[1], 1 / symb_rate / sps)
    cd = np.exp(1j * 2 * (np.pi * freq) ** 2 * tau_cd)
    H = _fiber_matrix(freq, tau_pmd, phiIQ, theta)
    out = np.zeros((2, rx.shape[1]), dtype=np.complex128)
    out[0] = (H[0][0] * spec[0] + H[0][1] * spec[1]) * cd
    out[1] = (H[1][0] * spec[0] + H[1][1] * spec[1]) * cd
    return np.complex64(np.fft.ifft(out, axis=1))


def simulate_channel(tx_up, h_pulse, h_channel):
    """Pulse shaping then the (optional) extra impulse response, both 'valid' (shared_funcs.py:56-63)."""
    n_out = tx_up.shape[1] - h_pulse.shape[0] - h_channel.shape[0] + 2
    out = np.zeros((tx_up.shape[0], n_out), dtype=np.complex64)
    for p in range(tx_up.shape[0]):
        out[p] = np.convolve(np.convolve(tx_up[p], h_pulse, mode="valid"), h_channel, mode="valid")
    return out


def generate_data_shaping(N, amps, SNR, h_channel, P, pol, symb_rate, sps, tau_cd, tau_pmd, phiIQ, theta, device,
                          rng=None, noise=None):
    """One frame of received samples + TX reference (shared_funcs.py:65-90).

    Returns (rx[pol,2,sps*N] float32, data[pol,2,N] float16, sigma_n).  ``rng``: numpy Generator for the symbol draw
    (default: fresh ``np.random.default_rng()`` like the reference); ``noise``: object with ``randn`` (default: the
    global ``np.random`` like the reference)."""
    rng = np.random.default_rng() if rng is None else rng
    noise = np.random if noise is None else noise
    T, Mc = PULSE_SPAN, len(h_channel)
    N_conv = N + Mc + 4 * T
    data = rng.choice(amps, (pol * 2, N_conv), p=P)
    tx_up = np.zeros((pol, sps * (N_conv - 1) + 1), dtype=np.complex64)
    tx_up[:, ::sps] = data[0::pol, :] + 1j * data[1::pol, :]
    sig = simulate_channel(tx_up, rrcfir(T, sps, ROLL_OFF), h_channel)
    sig = simulate_dispersion(sig, symb_rate, sps, tau_cd, tau_pmd, phiIQ, theta)
    sigma_n = np.sqrt(np.mean(np.abs(sig) ** 2) * sps / 2 / 10 ** (SNR / 10))
    sig += sigma_n * (noise.randn(*sig.shape) + 1j * noise.randn(*sig.shape))
    rx = np.stack([sig[:, :sps * N].real, sig[:, :sps * N].imag], axis=1)
    lo = T + Mc - 1
    ref = np.stack([data[0::pol, lo:lo + N], data[1::pol, lo:lo + N]], axis=1)
    return (torch.from_numpy(np.ascontiguousarray(rx)).to(device, torch.float32),
            torch.from_numpy(np.ascontiguousarray(ref)).to(device, torch.float16), sigma_n)


def generate_data(N, M, amps, SNR, h_channel, sps, device, P, rng=None, noise=None):
    """Single-polarisation AWGN/ISI channel (AWGN_channel/func_VAELE_MQAM_shaping.py:39-61) -> (rx[2,sps*N], data[2,N])."""
    rng = np.random.default_rng() if rng is None else rng
    noise = np.random if noise is None else noise
    T = PULSE_SPAN
    N_conv = N + len(h_channel) + 4 * T
    data = rng.choice(amps, (2, N_conv), p=P)
    tx_up = np.zeros(sps * (N_conv - 1) + 1, dtype=np.complex64)
    tx_up[::sps] = data[0] + 1j * data[1]
    sig = np.convolve(np.convolve(tx_up, rrcfir(T, sps, ROLL_OFF), mode="valid"), h_channel, mode="valid")
    sigma_n = np.sqrt(sps * np.mean(np.abs(sig) ** 2) / 2 / 10 ** (SNR / 10))
    sig += sigma_n * (noise.randn(*sig.shape) + 1j * noise.randn(*sig.shape))
    rx = np.stack([sig[:sps * N].real, sig[:sps * N].imag])
    lo = T + M - 1
    ref = np.stack([data[0, lo:lo + N], data[1, lo:lo + N]])
    return (torch.from_numpy(np.ascontiguousarray(rx)).to(device, torch.float32),
            torch.from_numpy(np.ascontiguousarray(ref)).to(device, torch.float16))


class SeededStreams:
    """Reproducible stand-in for the reference's unseeded RNG use, frame by frame.

    Frame k draws its symbols from ``np.random.default_rng(seed + k)`` and all noise comes from ONE legacy
    ``RandomState(seed)`` stream -- the consumption pattern tools/capture_golden.py imposes on the reference, so the
    same seed yields the same frames here and there."""

    def __init__(self, seed):
        self.seed, self.k = int(seed), 0
        self.noise = np.random.RandomState(self.seed)

    def next_rng(self):
        g = np.random.default_rng(self.seed + self.k)
        self.k += 1
        return g


# ------------------------------------------------------------------ batched device generator (row f1)
def generate_batch_gpu(R, N, amps, P, SNR, h_channel, symb_rate, sps, tau_cd, tau_pmd, phiIQ, theta, device, generator=None):
    """The DP channel model for R runs at once with torch ops on ``device``.

    amps[n]; P[R,n] or [n]; SNR / theta: scalar or [R].  Returns (rx[R,2,2,sps*N] f32, data[R,2,2,N] f16).
    Same physics as generate_data_shaping (PCS draw, zero-stuffing, RRC, extra IR, CD+PMD+rotation+IQ phase, AWGN)."""
    dev = torch.device(device)
    T = PULSE_SPAN
    hc = torch.as_tensor(np.asarray(h_channel), dtype=torch.complex64, device=dev)
    Mc = hc.numel()
    N_conv = N + Mc + 4 * T
    amps_t = torch.as_tensor(np.asarray(amps), dtype=torch.float32, device=dev)
    Pt = torch.as_tensor(np.asarray(P), dtype=torch.float32, device=dev)
    if Pt.dim() == 1:
        Pt = Pt.expand(R, -1)
    idx = torch.multinomial(Pt, 4 * N_conv, replacement=True, generator=generator).reshape(R, 4, N_conv)
    data = amps_t[idx]                                                     # [R, 4, N_conv]: rows I0,Q0,I1,Q1
    sym = torch.complex(data[:, 0::2], data[:, 1::2])                      # [R, 2, N_conv]
    Lup = sps * (N_conv - 1) + 1
    up = torch.zeros(R, 2, Lup, dtype=torch.complex64, device=dev)
    up[:, :, ::sps] = sym
    hp = torch.as_tensor(rrcfir(T, sps, ROLL_OFF), dtype=torch.float32, device=dev)

    def conv_valid(x, k):                                                  # true convolution, 'valid'
        kk = torch.flip(k, [0]).reshape(1, 1, -1)
        xr = torch.nn.functional.conv1d(x.real.reshape(-1, 1, x.shape[-1]), kk.real if kk.is_complex() else kk)
        xi = torch.nn.functional.conv1d(x.imag.reshape(-1, 1, x.shape[-1]), kk.real if kk.is_complex() else kk)
        if kk.is_complex():
            yr = xr - torch.nn.functional.conv1d(x.imag.reshape(-1, 1, x.shape[-1]), kk.imag)
            yi = xi + torch.nn.functional.conv1d(x.real.reshape(-1, 1, x.shape[-1]), kk.imag)
            xr, xi = yr, yi
        return torch.complex(xr, xi).reshape(*x.shape[:-1], -1)

    sig = conv_valid(up, hp)
    if Mc > 1:
        sig = conv_valid(sig, hc)
    Ls = sig.shape[-1]
    spec = torch.fft.fft(sig.to(torch.complex128), dim=-1)
    freq = torch.fft.fftfreq(Ls, 1 / symb_rate / sps, device=dev, dtype=torch.float64)
    cd = torch.exp(1j * 2 * (math.pi * freq) ** 2 * tau_cd)
    d = torch.exp(1j * math.pi * tau_pmd * freq)
    th = torch.as_tensor(theta, dtype=torch.float64, device=dev).expand(R).reshape(R, 1)
    c, s = torch.cos(th), torch.sin(th)
    e = np.exp(-1j * np.asarray(phiIQ, dtype=np.complex128))
    e0, e1 = complex(e[0]), complex(e[1])
    di = 1 / d
    H00 = (c * e0) * d * (c * e0) + (-s * e0) * di * (-s * e1)
    H01 = (c * e0) * d * (s * e0) + (-s * e0) * di * (c * e1)
    H10 = (s * e1) * d * (c * e0) + (c * e1) * di * (-s * e1)
    H11 = (s * e1) * d * (s * e0) + (c * e1) * di * (c * e1)
    out = torch.stack([(H00 * spec[:, 0] + H01 * spec[:, 1]) * cd, (H10 * spec[:, 0] + H11 * spec[:, 1]) * cd], dim=1)
    sig = torch.fft.ifft(out, dim=-1).to(torch.complex64)
    snr = torch.as_tensor(SNR, dtype=torch.float32, device=dev).expand(R).reshape(R, 1, 1)
    sigma = torch.sqrt(sig.abs().square().mean(dim=(1, 2), keepdim=True) * sps / 2 / 10 ** (snr / 10))
    noise = torch.randn(R, 2, Ls, 2, device=dev, generator=generator)
    sig = sig + sigma * torch.complex(noise[..., 0], noise[..., 1])
    rx = torch.stack([sig.real[..., :sps * N], sig.imag[..., :sps * N]], dim=2).contiguous()   # [R,2,2,sps*N]
    lo = T + Mc - 1
    ref = torch.stack([data[:, 0::2, lo:lo + N], data[:, 1::2, lo:lo + N]], dim=2).to(torch.float16).contiguous()
    return rx.to(torch.float32), ref


# ------------------------------------------------------------------ HIP generator (vaeq_gen_dp_*): row f1
_GEO = {}


def _geo_cached(kind, fn, N, h_channel, sps):
    """The frame geometry depends on (N, impulse response, sps) only and is asked for once per frame / epoch (do not modify the returned arrays)."""
    key = (kind, int(N), np.asarray(h_channel).astype(np.complex64).tobytes(), int(sps))
    g = _GEO.get(key)
    if g is None:
        if len(_GEO) > 64:
            _GEO.clear()
        g = _GEO[key] = fn(N, h_channel, sps)
    return g


def dp_frame_geometry(N, h_channel, sps):
    """Lengths of the reference's generator chain (shared_funcs.py:66-73, 56-58, 89) and the combined 'valid' FIR g = pulse * IR."""
    return _geo_cached("dp", _dp_frame_geometry, N, h_channel, sps)


def _dp_frame_geometry(N, h_channel, sps):
    T = PULSE_SPAN
    hp = rrcfir(T, sps, ROLL_OFF)
    hc = np.asarray(h_channel).astype(np.complex64)
    g = np.convolve(hp.astype(np.complex128), hc.astype(np.complex128)).astype(np.complex64)
    Lc, Lg = len(hc), len(hp) + len(hc) - 1
    N_conv = N + Lc + 4 * T
    Ls = sps * (N_conv - 1) + 1 - Lg + 1
    return dict(g=g, Lg=Lg, N_conv=N_conv, Ls=Ls, ref_offset=T + Lc - 1)


_DEV_CONST = {}


def _dev_const(arr, dtype, device):
    """Device copy of a small host constant, cached by content: the generators are called once per frame / epoch with the same
    tables, and every fresh H2D copy of pageable memory is a host-side synchronisation point."""
    a = np.array(arr, copy=True, order="C")
    key = (a.tobytes(), a.shape, str(a.dtype), str(dtype), str(device))
    t = _DEV_CONST.get(key)
    if t is None:
        if len(_DEV_CONST) > 256:
            _DEV_CONST.clear()
        t = torch.as_tensor(a, device=device).to(dtype).contiguous()
        _DEV_CONST[key] = t
    return t


try:
    from xxhash import xxh3_64_intdigest as _fast_hash           # ~10 GB/s: a [8192, 8] probability table in 40 us
except ImportError:                                              # pragma: no cover
    def _fast_hash(b):
        return hash(bytes(b))

_CDF_DEV = {}


def _cdf_dev(P, R, n, device):
    """Device copy of the runs' cumulative PCS distributions [R, n] (float32), cached by the CONTENT of P: the generators are called once per
    frame / epoch with the same table, and tiling + cumulating + uploading it each time costs more host time than a short AWGN epoch takes on
    the GPU.  P: [n] (all runs alike) or [R, n]."""
    Pn = np.ascontiguousarray(P, dtype=np.float64)
    if Pn.shape not in ((n,), (R, n)):
        raise ValueError(f"P must have shape ({n},) or ({R}, {n}), got {Pn.shape}")
    key = (_fast_hash(memoryview(Pn).cast("B")), Pn.shape, R, str(device))
    t = _CDF_DEV.get(key)
    if t is None:
        if len(_CDF_DEV) > 64:
            _CDF_DEV.clear()
        c = np.cumsum(Pn, axis=-1).astype(np.float32)
        t = torch.as_tensor(c, device=device)
        t = (t.expand(R, n) if c.ndim == 1 else t).contiguous()
        _CDF_DEV[key] = t
    return t


def fast_fft_len(n):
    """Smallest m * 2^a >= n with m in {1, 3, 5}: lengths hipFFT runs as one radix-2/4/8-dominated kernel chain (measured on MI355X
    for [2048, 2, L] c2c: L = 20480 takes 1.6 ms per fft+ifft, 20250 = 2*3^4*5^3 3.3 ms, the Bluestein length 20034 6.6 ms); the multiples of
    1024 among them up to 20 * 1024 (4, 5, 8, 10, 16, 20) are the rows vaeq_gen_dp_frame transforms itself (csrc/vaeq_gen_fused.h)."""
    best = None
    for m in (1, 3, 5):
        v = m
        while v < n:
            v *= 2
        best = v if best is None else min(best, v)
    return best


STREAM_BLOCK = 8192   # runs per vaeq_gen_dp_frame call.  Part of the DEFINITION of the random streams (run r draws from the Philox key of block
                      # r // STREAM_BLOCK with run counter r % STREAM_BLOCK), hence a constant and not a tuning parameter: frames of a seed never
                      # depend on how much workspace a caller wants to spend (8192 runs = a 2.7 GB workspace for the default frame)


FUSED_ROWS = (4096, 5120, 8192, 10240, 16384, 20480)   # N1 * 1024, N1 in {4, 5, 8, 10, 16, 20}: the rows vaeq_gen_dp_frame transforms itself


def padded_row_len(n):
    """Row length of the "padded" DP frame: the smallest row the library's own three-pass form covers (FUSED_ROWS) when there is one -- for
    frames up to ~10 000 symbols no hipFFT plan is ever created (3.6 s on a process's first call) and no five-pass chain runs, at the price of a
    longer transform for lengths between the supported ones (N = 3000: 8192 instead of 6144) --, else the next {1,3,5} * 2^a length (hipFFT)."""
    for L in FUSED_ROWS:
        if L >= n:
            return L
    return fast_fft_len(n)


def generate_batch_hip(R, N, amps, P, SNR, h_channel, symb_rate, sps, tau_cd, tau_pmd, phiIQ, theta, device, seed, frame,
                       return_sigma=False, fft="padded"):
    """The DP channel model for R runs on the device: one vaeq_gen_dp_frame call per STREAM_BLOCK runs (the padded default frame takes the
    library's three-pass form, other row lengths the stage kernels around in-place hipFFT).

    fft: "exact"  -- dispersion applied on the FFT of the exact sequence length Ls like the reference (circular filtering; Ls = 20034 =
                     2*3^3*7*53 for the default frame costs hipFFT 4x the time of a 20480-point transform);
         "padded" -- rows zero-padded to padded_row_len(Ls + 64) (linear filtering; the library's own split-FFT rows wherever one fits): the
                     dispersion's impulse response spans a few samples, so only samples that close to the frame edges differ from "exact".
    Deterministic in (seed, frame, run): counter-based Philox streams.  Returns (rx[R,2,2,sps*N] f32, data[R,2,2,N] f16[, sigma_n[R]])."""
    import ctypes as C

    from . import _native as nat
    dev = torch.device(device)
    geo = dp_frame_geometry(N, h_channel, sps)
    n = len(amps)
    amp_t = _dev_const(amps, torch.float32, dev)
    cdf = _cdf_dev(P, R, n, dev)
    g_t = _dev_const(np.stack([geo["g"].real, geo["g"].imag], -1), torch.float32, dev)
    snr = _dev_const(np.broadcast_to(np.asarray(SNR, np.float32), (R,)), torch.float32, dev)
    th = _dev_const(np.broadcast_to(np.asarray(theta, np.float32), (R,)), torch.float32, dev)
    e = np.exp(-1j * np.asarray(phiIQ, dtype=np.complex128))
    rx = torch.empty(R, 2, 2, sps * N, dtype=torch.float32, device=dev)
    data = torch.empty(R, 2, 2, N, dtype=torch.float16, device=dev)
    sigma = torch.empty(R, dtype=torch.float32, device=dev)
    L = nat.lib()
    st = nat.current_stream(dev)
    if fft not in ("exact", "padded"):
        raise ValueError(f"fft must be 'exact' or 'padded', got {fft!r}")
    Lrow = geo["Ls"] if fft == "exact" else padded_row_len(geo["Ls"] + 64)
    with torch.cuda.device(dev):
        for r0 in range(0, R, STREAM_BLOCK):
            r1 = min(R, r0 + STREAM_BLOCK)
            Rc = r1 - r0
            sig = torch.empty(Rc, 2, Lrow, 2, dtype=torch.float32, device=dev)
            pw = torch.empty(Rc, L.vaeq_gen_dp_power_parts(Lrow), dtype=torch.float32, device=dev)   # the first pass's partial sums of |sig|^2
            # runs inside a block are told apart by the run counter word of the Philox streams, blocks by the key (_mix_seed)
            nat.check(L.vaeq_gen_dp_frame(Rc, N, geo["N_conv"], sps, n, geo["Lg"], geo["Ls"], Lrow, geo["ref_offset"], nat.ptr(amp_t),
                                          nat.ptr(cdf[r0:r1].contiguous()), nat.ptr(g_t), nat.ptr(snr[r0:r1].contiguous()),
                                          nat.ptr(th[r0:r1].contiguous()), float(symb_rate) * sps, float(tau_cd), float(tau_pmd),
                                          float(e[0].real), float(e[0].imag), float(e[1].real), float(e[1].imag),
                                          C.c_uint64(_mix_seed(seed, r0)), C.c_uint32(frame), nat.ptr(sig), nat.ptr(pw), nat.ptr(rx[r0:r1]),
                                          nat.ptr(data[r0:r1], torch.float16), nat.ptr(sigma[r0:r1]), st), "vaeq_gen_dp_frame")
    return (rx, data, sigma) if return_sigma else (rx, data)


def awgn_frame_geometry(N, h_channel, sps):
    """Lengths of generate_data (AWGN_channel/func_VAELE_MQAM_shaping.py:39-61): combined pulse g = rrc * h_channel, its 'valid'
    output length Ls and the offset of the TX reference (M_channel = number of symbol-spaced channel taps)."""
    return _geo_cached("awgn", _awgn_frame_geometry, N, h_channel, sps)


def _awgn_frame_geometry(N, h_channel, sps):
    T = PULSE_SPAN
    h = np.asarray(h_channel, dtype=np.complex64)
    M_channel = (len(h) - 1) // sps + 1
    N_conv = N + len(h) + 4 * T
    g = np.convolve(rrcfir(T, sps, ROLL_OFF).astype(np.complex64), h).astype(np.complex64)
    Ls = sps * (N_conv - 1) + 1 - len(g) + 1
    return dict(N_conv=N_conv, g=g, Lg=len(g), Ls=Ls, ref_offset=T + M_channel - 1)


def generate_awgn_batch_hip(R, N, amps, P, SNR, h_channel, sps, device, seed, frame, return_sigma=False, sigma_fixed=None):
    """The AWGN/ISI channel model for R runs on the device (vaeq_gen_awgn): deterministic in (seed, frame, run).

    amps[n]; P[R,n] or [n]; SNR scalar or [R]; sigma_fixed (scalar or [R]) replaces the power-derived noise level (the VAE-NN
    script's generate_data).  Returns (rx[R,2,sps*N] f32, data[R,2,N] f16[, sigma_n[R]])."""
    import ctypes as C

    from . import _native as nat
    dev = torch.device(device)
    geo = awgn_frame_geometry(N, h_channel, sps)
    n = len(amps)
    amp_t = _dev_const(amps, torch.float32, dev)
    cdf = _cdf_dev(P, R, n, dev)
    g_t = _dev_const(np.stack([geo["g"].real, geo["g"].imag], -1), torch.float32, dev)
    snr = _dev_const(np.broadcast_to(np.asarray(SNR, np.float32), (R,)), torch.float32, dev)
    rx = torch.empty(R, 2, sps * N, dtype=torch.float32, device=dev)
    data = torch.empty(R, 2, N, dtype=torch.float16, device=dev)
    sigma = torch.empty(R, dtype=torch.float32, device=dev)
    sig = torch.empty((R, geo["Ls"], 2) if sps != 2 else (1,), dtype=torch.float32, device=dev)   # sps == 2: the fused kernels need none
    pw = torch.empty(R, (geo["Ls"] + 2047) // 2048, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_gen_awgn(R, N, geo["N_conv"], sps, n, geo["Lg"], geo["Ls"], geo["ref_offset"], nat.ptr(amp_t), nat.ptr(cdf),
                                          nat.ptr(g_t), nat.ptr(snr), C.c_uint64(_mix_seed(seed, 0)), C.c_uint32(frame), nat.ptr(sig),
                                          nat.ptr(pw), nat.ptr(rx), nat.ptr(data, torch.float16), nat.ptr(sigma),
                                          None if sigma_fixed is None else
                                          nat.ptr(_dev_const(np.broadcast_to(np.asarray(sigma_fixed, np.float32), (R,)), torch.float32, dev)),
                                          nat.current_stream(dev)), "vaeq_gen_awgn")
    return (rx, data, sigma) if return_sigma else (rx, data)


class CleanAwgnFrame:
    """A frame of generate_awgn_clean_batch_hip: the noise-free channel output and what the consumer needs to add vaeq_gen_awgn's noise to it
    while reading (engine.AWGNEngine.validate_clean)."""
    __slots__ = ("sig", "power", "data", "snr", "sigma_fixed", "seed", "frame", "R", "N", "sps", "Ls")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def awgn_clean_supported(sps, M_est):
    """Shapes for which the validation pass can add the noise itself (vaeq_awgn_validate_gen: sps == 2, baked tap counts)."""
    return sps == 2 and M_est in (9, 17, 25)


def generate_awgn_clean_batch_hip(R, N, amps, P, SNR, h_channel, sps, device, seed, frame, sigma_fixed=None):
    """generate_awgn_batch_hip without its last stage (vaeq_gen_awgn_clean, sps == 2): symbols, pulse shaping and channel once; the noise of
    the same (seed, frame) is added by the kernel that reads the frame.  The reference's validation frame (func_VAELE_MQAM_shaping.py:310,
    N_valid = 15 000 fresh symbols per evaluated epoch) is read exactly once, so it never needs to exist in its noisy form."""
    import ctypes as C

    from . import _native as nat
    if sps != 2:
        raise ValueError("the clean-frame form of the AWGN generator exists for sps == 2 only")
    dev = torch.device(device)
    geo = awgn_frame_geometry(N, h_channel, sps)
    n = len(amps)
    amp_t = _dev_const(amps, torch.float32, dev)
    cdf = _cdf_dev(P, R, n, dev)
    g_t = _dev_const(np.stack([geo["g"].real, geo["g"].imag], -1), torch.float32, dev)
    snr = _dev_const(np.broadcast_to(np.asarray(SNR, np.float32), (R,)), torch.float32, dev)
    sig = torch.empty(R, geo["Ls"], 2, dtype=torch.float32, device=dev)
    data = torch.empty(R, 2, N, dtype=torch.float16, device=dev)
    pw = torch.empty(R, (geo["Ls"] + 2047) // 2048, dtype=torch.float32, device=dev)
    key = _mix_seed(seed, 0)
    with torch.cuda.device(dev):
        nat.check(nat.lib().vaeq_gen_awgn_clean(R, N, geo["N_conv"], sps, n, geo["Lg"], geo["Ls"], geo["ref_offset"], nat.ptr(amp_t), nat.ptr(cdf),
                                                nat.ptr(g_t), C.c_uint64(key), C.c_uint32(frame), nat.ptr(sig), nat.ptr(pw),
                                                nat.ptr(data, torch.float16), nat.current_stream(dev)), "vaeq_gen_awgn_clean")
    sf = None if sigma_fixed is None else _dev_const(np.broadcast_to(np.asarray(sigma_fixed, np.float32), (R,)), torch.float32, dev)
    return CleanAwgnFrame(sig=sig, power=pw, data=data, snr=snr, sigma_fixed=sf, seed=key, frame=int(frame), R=R, N=N, sps=sps, Ls=geo["Ls"])


def _mix_seed(seed, r0):
    """Key of the Philox streams of the chunk that starts at run r0 (runs inside a chunk are told apart by the run counter word)."""
    return (int(seed) * 0x9E3779B97F4A7C15 + int(r0) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
