"""Per-frame epilogue for a batch of runs, as torch ops on whatever device the tensors live on.

Restates, vectorised over a leading run axis, optical_DP_channel/shared_funcs.py:188-338 (find_shift,
find_shift_symb_full, SER_IQflip, SER_constell_shaping, dec_on_bound) and the roll / cut / slice logic of
func_VAELE_DP_MQAM_shaping.py:68-89 and func_VAEflex_DP_MQAM_shaping.py:72-84.  Per-run data-dependent rolls and
slices become index arithmetic and masks, so one call serves the whole sweep without host round trips.
(PyTorch here is plumbing around the HIP training kernel; fusing these reductions into a kernel is row f2.)
"""
import torch

N_SHIFT = 21   # lags searched (func_VAELE_DP_MQAM_shaping.py:70,81)
N_CUT = 10     # symbols dropped at the end of every minibatch (func_VAELE_DP_MQAM_shaping.py:40)
EDGE = 11      # symbols dropped at both frame ends (func_VAELE_DP_MQAM_shaping.py:79)


def shift_search(E, tx, n_shift=N_SHIFT):
    """E[R,2,N] (equaliser side, per output polarisation), tx[R,2,2,N] -> (shift[R,2] int64, r[R] int64).

    shared_funcs.py:299-314 / 322-338: |corr| of TX I and Q of both polarisations with E rolled by -10..10 symbols;
    the pairing (straight or swapped) with the larger summed peak wins."""
    half = n_shift // 2
    txf = tx.float()
    corr = []
    for i in range(n_shift):
        Er = torch.roll(E, i - half, dims=-1)                               # E_mat[..., i], :302
        corr.append(torch.einsum("racn,rbn->rcba", txf, Er))                # [R, c(I/Q), b(E pol), a(tx pol)]
    corr = torch.stack(corr, dim=-1).abs()
    cmax, cind = corr.max(dim=-1)                                           # over lags, :303-304
    cm, imax = cmax.max(dim=1)                                              # over I/Q, :305   -> [R,b,a]
    pick = torch.gather(cind, 1, imax.unsqueeze(1)).squeeze(1)              # corr_ind[ind_max[b,a], b, a]
    ind_XY = torch.stack([pick[:, 0, 0], pick[:, 1, 1]], dim=1)             # :307-309
    ind_YX = torch.stack([pick[:, 0, 1], pick[:, 1, 0]], dim=1)
    straight = (cm[:, 0, 0] + cm[:, 1, 1]) >= (cm[:, 0, 1] + cm[:, 1, 0])   # :311
    shift = half - torch.where(straight.unsqueeze(1), ind_XY, ind_YX)
    return shift, (~straight).long()


def _align(t, shift, r):
    """roll polarisation axis by r, then time axis of pol p by -shift[p] (func_VAELE_DP...:71-72): t[R,2,C,N]."""
    R, _, Cc, N = t.shape
    pol = (torch.arange(2, device=t.device).unsqueeze(0) - r.unsqueeze(1)) % 2           # [R,2] source pol
    t = torch.gather(t, 1, pol.reshape(R, 2, 1, 1).expand(R, 2, Cc, N))
    idx = (torch.arange(N, device=t.device).reshape(1, 1, N) + shift.unsqueeze(-1)) % N   # out[n] = in[n+shift]
    return torch.gather(t, 3, idx.unsqueeze(2).expand(R, 2, Cc, N))


def _keep_mask(shift, N, batch_len, device):
    """Boolean [R,N]: symbols that survive the per-minibatch cut (:73-77) and the frame-edge slice (:79)."""
    R = shift.shape[0]
    ms = shift.abs().amax(dim=1)                                             # max |shift|
    n = torch.arange(N, device=device).unsqueeze(0)
    if batch_len is None:                                                    # VAEflex: only [11 : -11-ms]
        return (n >= EDGE) & (n < (N - EDGE - ms).unsqueeze(1))
    Lk = (batch_len - shift[:, 0] - N_CUT).clamp(0, batch_len).unsqueeze(1)  # kept per minibatch
    m, j = n // batch_len, n % batch_len
    k = m * Lk + j                                                           # rank in the compacted sequence
    K = (N // batch_len) * Lk
    return (j < Lk) & (k >= EDGE) & (k < K - EDGE - ms.unsqueeze(1))


def _levels(tx, n_lev):
    scale = (n_lev - 1) / 2
    data = torch.round(scale * tx.float() + scale)                           # :198 / :239
    inv = torch.stack([data[:, :, 0], -(data[:, :, 1] - 2 * scale)], dim=2)  # :199 / :240
    return data, inv, scale


def _masked_rate(err, mask):
    """err[R,2,N] bool, mask[R,N] -> error rate per (run, pol) over the kept symbols."""
    cnt = mask.sum(dim=1).clamp(min=1).unsqueeze(1).float()
    return (err & mask.unsqueeze(1)).sum(dim=-1).float() / cnt


def ser_soft_demap(dec, tx, mask, n_lev):
    """SER_IQflip (shared_funcs.py:188-222) on hard decisions dec[R,2,2,N] = argmax(q) per axis, masked."""
    data, inv, scale = _levels(tx, n_lev)
    dec = dec.float()
    dec_pi = -(dec - 2 * scale)                                              # :206
    dec_pi4 = torch.stack([-(dec[:, :, 1] - 2 * scale), dec[:, :, 0]], dim=2)  # :212
    dec_3pi4 = -(dec_pi4 - 2 * scale)                                        # :217
    rates = []
    for d in (dec, dec_pi, dec_pi4, dec_3pi4):
        rates.append(_masked_rate(((data - d) != 0).any(dim=2), mask))
        rates.append(_masked_rate(((inv - d) != 0).any(dim=2), mask))
    return torch.stack(rates, dim=0).amin(dim=0)                             # :221  -> [R,2]


def ser_constellation(y, tx, mask, amp, nu_sc, var0, return_scale=False):
    """SER_constell_shaping + dec_on_bound (shared_funcs.py:225-287) on aligned FIR outputs y[R,2,2,N], masked.
    return_scale: also return the mean-radius normalisation factor [R] of :242 (the reference applies it IN PLACE to its argument)."""
    n_lev = amp.numel()
    R = y.shape[0]
    d_vec = (1 + 2 * nu_sc * var0).reshape(R, 1) * ((amp[:-1] + amp[1:]) / 2).reshape(1, -1)   # :234
    inf = torch.full((R, 1), float("inf"), device=y.device)
    lo, hi = torch.cat([-inf, d_vec], dim=1), torch.cat([d_vec, inf], dim=1)                     # :235-236
    data, inv, scale = _levels(tx, n_lev)
    txf = tx.float()
    mk = mask.unsqueeze(1).float()
    cnt = (2 * mask.sum(dim=1)).clamp(min=1).float()
    num = (torch.sqrt(txf[:, :, 0] ** 2 + txf[:, :, 1] ** 2) * mk).sum(dim=(1, 2)) / cnt
    den = (torch.sqrt(y[:, :, 0] ** 2 + y[:, :, 1] ** 2) * mk).sum(dim=(1, 2)) / cnt
    scale_n = num / den
    y = y * scale_n.reshape(R, 1, 1, 1)                                                          # :242

    def on_bound(r, d):                                                                          # :267-287
        di = d.long().clamp(0, n_lev - 1).reshape(R, -1)
        l = torch.gather(lo, 1, di).reshape(d.shape)
        h = torch.gather(hi, 1, di).reshape(d.shape)
        ok = (l <= r) & (r < h)
        return _masked_rate(~(ok[:, :, 0] & ok[:, :, 1]), mask)

    y_pi4 = torch.stack([-y[:, :, 1], y[:, :, 0]], dim=2)                                        # :255
    rates = []
    for r_ in (y, -y, y_pi4, -y_pi4):                                                            # :245-262
        rates.append(on_bound(r_, data))
        rates.append(on_bound(r_, inv))
    ser = torch.stack(rates, dim=0).amin(dim=0)                                                  # :264
    return (ser, scale_n) if return_scale else ser


def dp_frame_epilogue(q, y, data, amp, nu_sc, var, batch_len=None):
    """Whole per-frame epilogue for R runs.

    q[R,2,2n,N] (out_train), y[R,2,2,N] (out_const), data[R,2,2,N] fp16 TX reference, amp[n], nu_sc[R], var[R,2].
    batch_len: minibatch length for the VAE-LE per-minibatch cut, None for VAEflex.
    Returns dict: SER[R,4] (const x, const y, soft x, soft y -- the row order of SER_valid, :79,89),
    shift_q/shift_c [R,2], r_q/r_c [R]."""
    n_lev = amp.numel()
    N = q.shape[-1]
    Eq = torch.einsum("i,rpin->rpn", amp, q[:, :, :n_lev])                                       # :296-297
    shift_q, r_q = shift_search(Eq, data)
    dec = torch.stack([q[:, :, :n_lev].argmax(dim=2), q[:, :, n_lev:].argmax(dim=2)], dim=2)     # :201
    dec = _align(dec, shift_q, r_q)
    ser_q = ser_soft_demap(dec, data, _keep_mask(shift_q, N, batch_len, q.device), n_lev)
    shift_c, r_c = shift_search(y[:, :, 0], data)
    ya = _align(y, shift_c, r_c)
    ser_c = ser_constellation(ya, data, _keep_mask(shift_c, N, batch_len, q.device), amp, nu_sc, var[:, 0])
    return dict(SER=torch.cat([ser_c, ser_q], dim=1), shift_q=shift_q, r_q=r_q, shift_c=shift_c, r_c=r_c)
