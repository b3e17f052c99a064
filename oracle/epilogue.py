"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the per-frame epilogue.

Citations: DP = optical_DP_channel/shared_funcs.py, LEDP = func_VAELE_DP_MQAM_shaping.py,
FLEX = func_VAEflex_DP_MQAM_shaping.py, AWGN = AWGN_channel/func_VAELE_MQAM_shaping.py.
Pinned by tests/golden/G5_dp_epilogue.npz and G7_runs.npz; cma_frame_epilogue by G14_cma_epilogue_*.npz.
"""
import numpy as np


def _shift_from_corr(E, tx, N_shift):
    """Common tail of find_shift / find_shift_symb_full (DP:299-314 / 322-338).

    E[2,N]: equaliser-side sequence per output polarisation; tx[2,2,N]."""
    N = E.shape[-1]
    half = N_shift // 2
    E_mat = np.stack([np.roll(E, i - half, axis=-1) for i in range(N_shift)], axis=-1)   # [b, n, i]   DP:300-302
    corr_max = np.empty((2, 2, 2), np.float32)
    corr_ind = np.empty((2, 2, 2), np.int64)
    for c in range(2):                                                                    # DP:303-304
        cc = np.abs(np.einsum("an,bni->bai", tx[:, c, :].astype(np.float32), E_mat.astype(np.float32)))
        corr_max[c], corr_ind[c] = cc.max(-1), cc.argmax(-1)
    ind_max = corr_max.argmax(0)                                                          # DP:305
    cm = corr_max.max(0)
    ind_XY = np.array([corr_ind[ind_max[0, 0], 0, 0], corr_ind[ind_max[1, 1], 1, 1]])     # DP:307-309
    ind_YX = np.array([corr_ind[ind_max[0, 1], 0, 1], corr_ind[ind_max[1, 0], 1, 0]])
    if cm[0, 0] + cm[1, 1] >= cm[0, 1] + cm[1, 0]:                                        # DP:311-314
        return half - ind_XY, 0
    return half - ind_YX, 1


def find_shift(q, tx, N_shift, amp_levels):
    """DP:290-314: correlate E_q[x_I] with TX I/Q of both polarisations over N_shift lags."""
    n = q.shape[1] // 2
    E = np.einsum("i,pin->pn", amp_levels.astype(np.float32), q[:, :n, :].astype(np.float32))   # DP:296-297
    return _shift_from_corr(E, tx, N_shift)


def find_shift_symb_full(rx, tx, N_shift):
    """DP:316-338: same with the FIR output's in-phase component."""
    return _shift_from_corr(rx[:, 0, :], tx, N_shift)


def _tx_levels(tx, n):
    scale = (n - 1) / 2
    data = np.round(scale * tx.astype(np.float32) + scale)                                # DP:198 / 239
    inv = data.copy()
    inv[:, 1, :] = -(data[:, 1, :] - scale * 2)                                           # DP:199 / 240
    return data, inv, scale


def SER_IQflip(q, tx):
    """DP:188-222: SER from argmax(q), min over IQ-flip x {0, pi, pi/2, 3pi/2}."""
    n = q.shape[1] // 2
    data, inv, scale = _tx_levels(tx, n)
    dec = np.stack([q[:, :n, :].argmax(1), q[:, n:, :].argmax(1)], axis=1).astype(np.float32)   # DP:201
    dec_pi = -(dec - scale * 2)                                                           # DP:206
    dec_pi4 = np.stack([-(dec[:, 1, :] - scale * 2), dec[:, 0, :]], axis=1)               # DP:212
    dec_3pi4 = -(dec_pi4 - scale * 2)                                                     # DP:217
    SER = np.ones((2, 2, 4), np.float32)
    for k, d in enumerate((dec, dec_pi, dec_pi4, dec_3pi4)):
        SER[0, :, k] = ((data - d) != 0).any(1).astype(np.float32).mean(-1)
        SER[1, :, k] = ((inv - d) != 0).any(1).astype(np.float32).mean(-1)
    return SER.min(axis=(0, 2))                                                           # DP:221


def SER_constell_shaping(rx, tx, amp_levels, nu_sc, var, inplace=None):
    """DP:225-287: SER from the FIR output against PCS-aware decision thresholds.  The reference rescales its `rx` argument in place
    (DP:242, `rx *= ...`); callers that hand in a slice view see that (the CMA modules do): pass the view as `inplace` to get it."""
    n = amp_levels.shape[0]
    a = amp_levels.astype(np.float32)
    d_vec = (np.float32(1) + np.float32(2 * nu_sc) * np.float32(var[0])) * (a[:-1] + a[1:]) / 2    # DP:234
    lo = np.concatenate(([-np.inf], d_vec)).astype(np.float32)                            # DP:235
    hi = np.concatenate((d_vec, [np.inf])).astype(np.float32)                             # DP:236
    data, inv, scale = _tx_levels(tx, n)
    data, inv = data.astype(np.int64), inv.astype(np.int64)
    txf = tx.astype(np.float32)
    rx = rx.astype(np.float32) * (np.mean(np.sqrt(txf[:, 0] ** 2 + txf[:, 1] ** 2, dtype=np.float32), dtype=np.float32)
                                  / np.mean(np.sqrt(rx[:, 0] ** 2 + rx[:, 1] ** 2, dtype=np.float32), dtype=np.float32))  # DP:242
    if inplace is not None:
        inplace[...] = rx

    def on_bound(r, d):                                                                   # DP:267-287
        ok = (lo[d] <= r) & (r < hi[d])
        return (~(ok[:, 0] & ok[:, 1])).astype(np.float32).mean(-1)

    rx_pi4 = np.stack([-rx[:, 1], rx[:, 0]], axis=1)                                      # DP:255
    SER = np.ones((2, 2, 4), np.float32)
    for k, r in enumerate((rx, -rx, rx_pi4, -rx_pi4)):                                    # DP:245-262
        SER[0, :, k] = on_bound(r, data)
        SER[1, :, k] = on_bound(r, inv)
    return SER.min(axis=(0, 2))                                                           # DP:264


def _align(t, shift, r):
    t = np.roll(t, r, axis=0)                                                             # LEDP:71 / 82
    out = t.copy()
    out[0], out[1] = np.roll(t[0], -int(shift[0]), axis=-1), np.roll(t[1], -int(shift[1]), axis=-1)   # LEDP:72 / 83
    return out


def dp_frame_epilogue(out_train, out_const, data, amp_levels, nu_sc, var, batch_len=None, N_cut=10, N_shift=21):
    """LEDP:70-89 (batch_len given: per-minibatch edge cut) or FLEX:74-84 (batch_len None: no cut).

    Returns dict(SER[4] = const x/y then soft-demap x/y, shift_q, r_q, shift_c, r_c)."""
    res = {}
    SER = np.empty(4, np.float32)
    for kind, seq in (("q", out_train), ("c", out_const)):
        if kind == "q":
            shift, r = find_shift(seq, data, N_shift, amp_levels)                         # LEDP:70
        else:
            shift, r = find_shift_symb_full(seq, data, N_shift)                           # LEDP:81
        al, dt = _align(seq, shift, r), data
        if batch_len is not None:                                                         # LEDP:73-77 / 84-87
            pol, rows, N = al.shape
            keep = slice(None, batch_len - int(shift[0]) - N_cut)
            al = al.reshape(pol, rows, N // batch_len, batch_len)[:, :, :, keep].reshape(pol, rows, -1)
            dt = data.reshape(pol, 2, N // batch_len, batch_len)[:, :, :, keep].reshape(pol, 2, -1)
        ms = int(np.max(np.abs(shift)))
        sl = slice(11, -11 - ms)                                                          # LEDP:79 / 89
        if kind == "q":
            SER[2:] = SER_IQflip(al[:, :, sl], dt[:, :, sl])
        else:
            SER[:2] = SER_constell_shaping(al[:, :, sl].copy(), dt[:, :, sl], amp_levels, nu_sc, var)
        res["shift_" + kind], res["r_" + kind] = shift, r
    res["SER"] = SER
    return res


# ------------------------------------------------------------------ AWGN validation (AWGN:188-204, 97-123)
def awgn_find_shift(q, tx, N_shift, amp_levels):
    n = amp_levels.shape[0]
    E = np.einsum("i,in->n", amp_levels.astype(np.float32), q[:n, :1000].astype(np.float32))   # AWGN:189-190
    half = N_shift // 2
    E_mat = np.stack([np.roll(E, i - half) for i in range(N_shift)], axis=-1)                 # AWGN:194-195
    corr = tx[0, :1000].astype(np.float32) @ E_mat                                            # AWGN:196
    if np.max(np.abs(corr)) >= 0.02 * q.shape[-1]:                                            # AWGN:197-198
        return half - int(np.argmax(np.abs(corr)))
    corr_IQ = tx[1, :1000].astype(np.float32) @ E_mat                                         # AWGN:200-204
    if np.max(np.abs(corr_IQ)) >= np.max(np.abs(corr)):
        return half - int(np.argmax(np.abs(corr_IQ)))
    return half - int(np.argmax(np.abs(corr)))


def awgn_SER_q(q, tx, n):
    """AWGN:97-123."""
    N = tx.shape[-1]
    scale = (n - 1) / 2
    data = np.round(scale * tx.astype(np.float32) + scale)
    dec = np.stack([q[:n, :N].argmax(0), q[n:, :N].argmax(0)]).astype(np.float32)
    dec_pi = -(dec - scale * 2)
    dec_pi4 = np.stack([-(dec[1] - scale * 2), dec[0]])
    dec_3pi4 = -(dec_pi4 - scale * 2)
    return min(((data - d) != 0).any(0).astype(np.float32).mean() for d in (dec, dec_pi, dec_pi4, dec_3pi4))


def awgn_validate(q, data, amp_levels):
    """AWGN:317-318."""
    shift = awgn_find_shift(q, data, 21, amp_levels)
    n = amp_levels.shape[0]
    return awgn_SER_q(q[:, 11 + shift:-11], data[:, 11:-11 - shift], n), shift


def cpe(y, M_ma=501):
    """Viterbi-Viterbi carrier phase estimation (shared_funcs.py:139-186): y[2,2,N] -> phase-corrected y.  4th power, moving
    average over M_ma symbols (zero padded), atan2 / 4, unwrapping of pi/2 jumps, de-rotation."""
    y = np.asarray(y, dtype=np.float64)
    out = np.zeros_like(y)
    for p in range(2):
        a, b = y[p, 0], y[p, 1]
        a2, b2 = a * a, b * b
        p4r = a2 * a2 - 6 * a2 * b2 + b2 * b2
        p4i = 4 * (a2 * a * b - a * b2 * b)
        ker = np.full(M_ma, 1.0 / M_ma)
        mr, mi = np.convolve(p4r, ker, mode="same"), np.convolve(p4i, ker, mode="same")
        phi = np.arctan2(mi, -mr) / 4
        d = np.diff(phi)
        corr = np.concatenate([[0.0], np.cumsum((d < -np.pi / 4).astype(float) - (d > np.pi / 4).astype(float))]) * (np.pi / 2)
        phi = phi + corr
        c, s = np.cos(phi), np.sin(phi)
        out[p, 0], out[p, 1] = a * c - b * s, b * c + a * s
    return out


def cma_frame_epilogue(out_const, data, amp_levels, nu_sc, var, soft_dec, N_cut=10, N_shift=21):
    """The two-stage epilogue of the constant-modulus modules (func_CMA_DP_MQAM_shaping.py:39-53, same in CMAbatch :37-51 / CMAflex):
    out_const[2,2,K] = equaliser output of one frame, data[2,2,K]; soft_dec(out, var, amp, nu_sc) = DP:529-542 (oracle.dp_soft_dec).

    :44 hands SER_constell_shaping a slice VIEW of out_const, whose in-place normalisation (DP:242) therefore rescales the kept window
    [11 : -11 - max|shift|] of out_const itself -- the soft demapper of :48 sees the normalised constellation there and the raw scale
    at the frame edges."""
    y = cpe(out_const[:, :, N_cut:-N_cut]).astype(np.float32)                              # :39
    d = data[:, :, N_cut:-N_cut]                                                          # :40
    shift_c, r_c = find_shift_symb_full(y, d, N_shift)                                    # :41
    y = _align(y, shift_c, r_c)                                                           # :42-43
    sl = slice(11, -11 - int(np.max(np.abs(shift_c))))
    SER = np.empty(4, np.float32)
    SER[:2] = SER_constell_shaping(y[:, :, sl], d[:, :, sl], amp_levels, nu_sc, var, inplace=y[:, :, sl])   # :44
    q = soft_dec(y, var, amp_levels, nu_sc)                                               # :48
    shift_q, r_q = find_shift(q, d, N_shift, amp_levels)                                  # :49
    q = _align(q, shift_q, r_q)                                                           # :50-51
    sl = slice(11, -11 - int(np.max(np.abs(shift_q))))
    SER[2:] = SER_IQflip(q[:, :, sl], d[:, :, sl])                                        # :52
    return dict(SER=SER, shift_c=shift_c, r_c=r_c, shift_q=shift_q, r_q=r_q, y=y)
