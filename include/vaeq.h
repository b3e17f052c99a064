/*
 * vaeq.h -- C ABI of libvaeq_hip.so: the MI355X (gfx950) implementation of the
 * VAE blind-equalizer training inner loop of kit-cel/vae-equalizer.
 *
 * The reference has no FFI: its "operator interface" for this path is a set of
 * Python callables.  Each entry point below names the
 * reference code it replaces (paths relative to the reference tree) and is what
 * a binding of that code would call; INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to row-major fp32 unless noted;
 *   - the caller owns all buffers; kernels keep no state between calls
 *     (equalizer taps, channel estimate, Adam moments and step counters are
 *     caller-owned arrays that are updated in place);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     are asynchronous with respect to the host;
 *   - return value: 0 = ok, negative = error (vaeq_strerror), never throws;
 *   - one "run" = one independent Monte-Carlo run / sweep point
 *     (optical_DP_channel/Eval_run_DP.py:68-86); runs never communicate.
 */
#ifndef VAEQ_H
#define VAEQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAEQ_VERSION 100

enum {
    VAEQ_OK = 0,             /* also for an empty batch (R == 0): nothing is read, pointers may be NULL */
    VAEQ_ERR_NULL = -1,      /* a required pointer is NULL */
    VAEQ_ERR_SHAPE = -2,     /* inconsistent or unsupported sizes (even M, n_lev not in {2,4,8}, window past S ...) */
    VAEQ_ERR_LDS = -3,       /* the per-run working set does not fit the 160 KiB LDS of a CU */
    VAEQ_ERR_LAUNCH = -4,    /* HIP launch error (hipGetLastError) */
    VAEQ_ERR_DEVICE = -5     /* not a gfx950 device / no device */
};

/* ------------------------------------------------------------------------
 * Dual-polarisation VAE-LE / VAEflex training loop.
 *
 * Replaces, for R independent runs at once, the minibatch loop
 *   optical_DP_channel/func_VAELE_DP_MQAM_shaping.py:57-66   (VAE-LE)
 *   optical_DP_channel/func_VAEflex_DP_MQAM_shaping.py:59-70 (VAEflex)
 * i.e. per step: twoXtwoFIR.forward (shared_funcs.py:500-527), loss_function_shaping
 * (shared_funcs.py:92-137), loss.backward(), optim.Adam.step() for both parameter
 * groups (func_VAELE_DP_MQAM_shaping.py:28,31), and the copies into
 * out_train / out_const / var_est (:61-62,64).
 *
 * Window of step s of frame f starts at symbol s*stride_sym of that frame's row
 * and spans B symbols (= B*sps samples, zero-padded by M/2 samples on both sides
 * exactly like Conv1d(padding=M//2), shared_funcs.py:494).
 *   VAE-LE : stride_sym = B,         keep_off = 0,               keep_len = B
 *   VAEflex: stride_sym = flex_step, keep_off = (B-flex_step)/2, keep_len = flex_step
 */
typedef struct vaeq_dp_args {
    int32_t R;           /* independent runs in this call (one workgroup each) */
    int32_t n_frames;    /* frames per run held in rx (taps/Adam state carry across frames; a launch of up to 16 frames is bit-identical to
                            the same frames launched one by one: the Adam bias corrections restart at every frame head as at a launch) */
    int32_t steps;       /* minibatch steps per frame */
    int32_t B;           /* batch_len: symbols per minibatch window */
    int32_t sps;         /* samples per symbol (reference: 2) */
    int32_t M;           /* M_est: taps of the butterfly FIR and of h_est; odd, <= 63 */
    int32_t n_lev;       /* ASK levels per axis: 2, 4 or 8 (4-/16-/64-QAM) */
    int32_t stride_sym;  /* symbols between consecutive window starts */
    int32_t keep_off;    /* first window-local symbol copied to q_out / y_out */
    int32_t keep_len;    /* number of window-local symbols copied per step */
    int64_t S;           /* samples per (run, frame, pol, I/Q) row of rx */
    const float *rx;     /* [R][n_frames][2 pol][2 I/Q][S]   received samples (rx_tensor, shared_funcs.py:88) */
    float *W;            /* [R][2][4][M]     FIR weight, nn.Conv1d(4,2,M) layout (shared_funcs.py:494) */
    float *h;            /* [R][2][2][2][M]  h_est[chi][nu][re/im][tap] (shared_funcs.py:583-586) */
    float *adam_mW, *adam_vW;   /* [R][2][4][M]     exp_avg / exp_avg_sq of W */
    float *adam_mh, *adam_vh;   /* [R][2][2][2][M]  exp_avg / exp_avg_sq of h */
    int32_t *step;       /* [R] Adam step count (in: steps done so far; out: += n_frames*steps) */
    const float *amp;    /* [n_lev]    amp_levels, shared by all runs (shared_funcs.py:568) */
    const float *P;      /* [R][n_lev] PCS pmf of the levels (shared_funcs.py:572) */
    const float *var;    /* [R][2]     demapper noise variance per polarisation (shared_funcs.py:581) */
    const float *nu_sc;  /* [R]        rescaled shaping factor (shared_funcs.py:570) */
    const float *lr_W;   /* [R] learning rate of param group 0 (W) for this call (func_VAELE_DP...:45-46) */
    const float *lr_h;   /* [R] learning rate of param group 1 (h_est) */
    float *q_out;        /* nullable [R][n_frames][2][2*n_lev][steps*keep_len]  out_train */
    float *y_out;        /* nullable [R][n_frames][2][2][steps*keep_len]        out_const */
    float *loss;         /* nullable [R][n_frames][steps]                        ELBO per minibatch */
    float *var_est;      /* nullable [R][n_frames][2][steps]                     C/(N-Mh) per minibatch */
    float *eq_out;       /* nullable [R][n_frames][2][steps*keep_len]   E_q[x_I] per polarisation: all find_shift reads of q
                            (shared_funcs.py:296-297) */
    int8_t *dec_out;     /* nullable [R][n_frames][2][2][steps*keep_len] argmax_i q_i per axis: all SER_IQflip reads of q (:201).
                            With eq_out + dec_out the per-frame epilogue needs no q: q_out may be NULL (32 of the 44 floats a
                            DP symbol costs in HBM are the materialised q) */
    float *dbg_gW;       /* nullable [R][2][4][M]     gradient of the LAST step (parity tests) */
    float *dbg_gh;       /* nullable [R][2][2][2][M] */
    int32_t threads;     /* kernel choice: 0 = automatic: the wave-per-run fast path when the shape allows (sps = 2, B <= 1024 even or
                            odd, M one of 9 13 17 21 25 31), else the generic kernel with 256 threads per run.
                            1 = wave-per-run only (VAEQ_ERR_SHAPE if unsupported); 64 / 128 / 256 = generic kernel, that
                            many threads per run */
    int32_t no_update;   /* 1: skip the Adam update (forward + loss + gradients only) */
} vaeq_dp_args;

int vaeq_dp_train(const vaeq_dp_args *args, void *stream);

/* SURVEY 8(b)'s `vaeq_dp_step_debug`: ONE minibatch step per run that additionally dumps the step's gradients, for teacher-forced parity
 * tests against loss.backward() of the reference (func_VAELE_DP_MQAM_shaping.py:64-65: dL/dW as net.conv_w.weight.grad [2][4][M], dL/dh_est
 * [2][2][2][M]).  = vaeq_dp_train on the first window of the first frame (n_frames and steps taken as 1) with dbg_gW / dbg_gh pointed at
 * gW / gh; args->no_update chooses between "gradients only" and "gradients + the Adam step"; every other field as for vaeq_dp_train.
 * (vaeq_dp_train itself dumps the LAST step's gradients when args->dbg_gW / dbg_gh are set: same kernels, same values.) */
int vaeq_dp_step_debug(const vaeq_dp_args *args, float *gW, float *gh, void *stream);

/* LDS bytes one run (= one workgroup) needs for the given shape, or a negative error code. */
int64_t vaeq_dp_lds_bytes(int32_t B, int32_t sps, int32_t M, int32_t n_lev);

/* How many runs of this shape the current device keeps co-resident (occupancy x compute units) with the kernel that
 * `threads` selects (same meaning as vaeq_dp_args.threads; VAE-LE geometry assumed).  Sweeps sized to a multiple of it
 * have no partially filled last round.  Negative error code on failure. */
int64_t vaeq_dp_resident_runs(int32_t B, int32_t sps, int32_t M, int32_t n_lev, int32_t threads);

/* ------------------------------------------------------------------------
 * Stand-alone soft demapper:  shared_funcs.py:529-542 (soft_dec), the same
 * formula as the demapping half of twoXtwoFIR.forward (shared_funcs.py:521-523).
 * y[R][2][2][N] -> q[R][2][2*n_lev][N];  amp[n_lev]; var[R][2]; nu_sc[R].
 */
int vaeq_soft_demap(int32_t R, int64_t N, int32_t n_lev, const float *y, const float *amp, const float *var,
                    const float *nu_sc, float *q, void *stream);

/* ------------------------------------------------------------------------
 * Butterfly FIR + soft demapper without training (twoXtwoFIR.forward in eval
 * mode, shared_funcs.py:500-527) on one zero-padded block of N symbols per run:
 * x[R][2][2][N*sps], W[R][2][4][M] -> q[R][2][2*n_lev][N] (nullable), y[R][2][2][N].
 */
int vaeq_dp_forward(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *W,
                    const float *amp, const float *var, const float *nu_sc, float *q, float *y, void *stream);

/* ------------------------------------------------------------------------
 * ELBO of one minibatch per run, values only: shared_funcs.py:92-137 (loss_function_shaping).
 * q[R][2][2*n_lev][B], x[R][2][2][B*sps], h[R][2][2][2][M], P[R][n_lev] -> loss[R], var_est[R][2].
 */
int vaeq_dp_loss(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x,
                 const float *h, const float *amp, const float *P, float *loss, float *var_est, void *stream);

/* ------------------------------------------------------------------------
 * Backward passes of the two stand-alone operators (for torch.autograd.Function wrappers: a reference-style
 * `loss.backward(); optimizer.step()` loop on HIP kernels; the fused vaeq_dp_train does not use them).
 *   vaeq_dp_loss_bwd   : loss_function_shaping (shared_funcs.py:92-137): g_up[R] = upstream d/dloss ->
 *                        gq[R][2][2*n_lev][B] = dL/dq, gh[R][2][2][2][M] = dL/dh_est
 *   vaeq_dp_forward_bwd: twoXtwoFIR.forward (shared_funcs.py:500-527): gq = dL/dq, gy = dL/dout (nullable), with the forward's
 *                        q, y -> gW[R][2][4][M] = dL/dW (softmin backward, then the conv weight gradient)
 */
int vaeq_dp_loss_bwd(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x, const float *h,
                     const float *amp, const float *P, const float *g_up, float *gq, float *gh, void *stream);
int vaeq_dp_forward_bwd(int32_t R, int32_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *q, const float *y,
                        const float *gq, const float *gy, const float *amp, const float *var, float *gW, void *stream);

/* ------------------------------------------------------------------------
 * Single-polarisation (AWGN / ISI channel) VAE-LE training loop.
 *
 * Replaces, for R independent runs at once, the minibatch loop
 *   AWGN_channel/func_VAELE_MQAM_shaping.py:297-306
 * i.e. per step: twoFIR.forward (:214-231, with the mean-|y| normalisation), loss_function (:63-95),
 * loss.backward(), optim.Adam(amsgrad=True).step() for both parameter groups (:283-286).
 * Minibatches are contiguous and non-overlapping: step s uses symbols [s*B, (s+1)*B) of the row.
 */
typedef struct vaeq_awgn_args {
    int32_t R;           /* independent runs (one workgroup each) */
    int32_t steps;       /* minibatch steps in this call (N_train // batch_len per epoch, :297) */
    int32_t B;           /* batch_len */
    int32_t sps;         /* samples per symbol */
    int32_t M;           /* M_est: taps of the FIR and of h_est; odd, <= 63 */
    int32_t n_lev;       /* ASK levels per axis: 2, 4 or 8 */
    int64_t S;           /* samples per (run, I/Q) row of rx */
    const float *rx;     /* [R][2 I/Q][S]  (rx_tensor, :58) */
    float *W;            /* [R][1][2][M]   nn.Conv1d(2,1,M) weight (:209) */
    float *h;            /* [R][2][M]      h_est re/im (:278-280) */
    float *adam_mW, *adam_vW, *adam_xW;   /* [R][2][M] exp_avg, exp_avg_sq, max_exp_avg_sq of W */
    float *adam_mh, *adam_vh, *adam_xh;   /* [R][2][M] ... of h */
    int32_t *step;       /* [R] Adam step count */
    const float *amp;    /* [n_lev]    amp_levels (:260) */
    const float *P;      /* [R][n_lev] pmf of the levels (:264) */
    const float *amp_mean; /* [R]      mean |Re|,|Im| of the shaped constellation (:271) */
    const float *var;    /* [R]        demapper variance 10^(-SNR/10) (:272) */
    const float *lr;     /* [R]        learning rate (both groups, no schedule) */
    float *q_out;        /* nullable [R][2*n_lev][steps*B] */
    float *y_out;        /* nullable [R][2][steps*B]  un-normalised FIR output (:227,231) */
    float *loss;         /* nullable [R][steps] */
    float *dbg_gW;       /* nullable [R][2][M] gradient of the LAST step */
    float *dbg_gh;       /* nullable [R][2][M] */
    int32_t threads;     /* 0 = default, else 64 / 128 / 256 */
    int32_t no_update;   /* 1: skip the Adam update */
} vaeq_awgn_args;

int vaeq_awgn_train(const vaeq_awgn_args *args, void *stream);
int64_t vaeq_awgn_lds_bytes(int32_t B, int32_t sps, int32_t M, int32_t n_lev);

/* Stand-alone ELBO of the single-polarisation variants for a given q (values): func_VAELE_MQAM_shaping.loss_function (:63-95) with
 * P[R][n_lev], func_VAENN_MQAM.loss_function (:63-95, entropy instead of KL) with P == NULL.
 * q[R][2*n_lev][B], x[R][2][B*sps], h[R][2][M] -> loss[R]. */
int vaeq_awgn_loss(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x, const float *h,
                   const float *amp, const float *P, float *loss, void *stream);

/* Backward passes of the two stand-alone AWGN operators (for autograd wrappers, like vaeq_dp_loss_bwd / vaeq_dp_forward_bwd):
 *   vaeq_awgn_loss_bwd   : g_up[R] -> gq[R][2*n_lev][B] = dL/dq, gh[R][2][M] = dL/dh   (P nullable as in vaeq_awgn_loss)
 *   vaeq_awgn_forward_bwd: twoFIR.forward (:214-231): gq[R][2*n_lev][N] (+ nullable gy[R][2][N] on the un-normalised output)
 *                          -> gW[R][2][M]; the forward is recomputed from x and W. */
int vaeq_awgn_loss_bwd(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x, const float *h,
                       const float *amp, const float *P, const float *g_up, float *gq, float *gh, void *stream);
int vaeq_awgn_forward_bwd(int32_t R, int32_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *W, const float *amp,
                          const float *amp_mean, const float *var, const float *gq, const float *gy, float *gW, void *stream);

/* twoFIR.forward in eval mode (validation pass, func_VAELE_MQAM_shaping.py:311-313) on N symbols per run:
 * x[R][2][N*sps], W[R][2][M] -> q[R][2*n_lev][N] (nullable), y[R][2][N] (un-normalised). */
int vaeq_awgn_forward(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *W,
                      const float *amp, const float *amp_mean, const float *var, float *q, float *y, void *stream);

/* ------------------------------------------------------------------------
 * Per-frame epilogue of the DP runs: shift / polarisation-swap search and both SER estimators,
 *   shared_funcs.py:188-338 with the roll / cut / slice logic of func_VAELE_DP_MQAM_shaping.py:68-89 (batch_len > 0)
 *   or func_VAEflex_DP_MQAM_shaping.py:72-84 (batch_len = 0: no per-minibatch cut).
 * q[R][2][2*n_lev][N] (out_train), y[R][2][2][N] (out_const), tx_f16[R][2][2][N] IEEE half (data_tensor, shared_funcs.py:89),
 * var[R][2], nu_sc[R] -> ser[R][4] (rows: constellation x, y; soft demapper x, y), shift[R][2 path][2] (path 0 = q, 1 = y),
 * rflag[R][2].  workspace: vaeq_dp_epilogue_ws_bytes(R, N) bytes of device memory. */
int vaeq_dp_epilogue(int32_t R, int64_t N, int32_t n_lev, int32_t batch_len, const float *q, const float *y, const void *tx_f16,
                     const float *amp, const float *var, const float *nu_sc, float *ser, int32_t *shift, int32_t *rflag,
                     void *workspace, void *stream);
int64_t vaeq_dp_epilogue_ws_bytes(int32_t R, int64_t N);
/* Same epilogue fed by the training kernel's compact outputs instead of q: eq[R][2][N] (vaeq_dp_args.eq_out) and dec[R][2][2][N]
 * (dec_out) of ONE frame; results are bit-identical to vaeq_dp_epilogue on the q of the same call. */
int vaeq_dp_epilogue_compact(int32_t R, int64_t N, int32_t n_lev, int32_t batch_len, const float *eq, const int8_t *dec, const float *y,
                             const void *tx_f16, const float *amp, const float *var, const float *nu_sc, float *ser, int32_t *shift,
                             int32_t *r_flag, void *stream);

/* The two-stage epilogue of the constant-modulus baselines in one launch (optical_DP_channel/func_CMA_DP_MQAM_shaping.py:39-52 after the phase
 * estimation; the CMAbatch / CMAflex modules are identical there): the constellation stage FIRST (find_shift_symb_full on y, roll / cut,
 * SER_constell_shaping), whose mean-radius normalisation stays in the kept window of the aligned output (the reference normalises a slice view in place,
 * shared_funcs.py:242), then soft_dec (:48) on that and the soft-demapper stage (find_shift_symb_full on E_q[x_I], SER_IQflip on argmax q) -- q is never
 * materialised.  y[R][2][2][N]: phase-corrected output cut to [10:-10]; tx_f16[R][2][2][N]: TX reference cut likewise; ser[R][4] = constellation SER of
 * both polarisations, then soft-demapper SER; shift[R][2][2], rflag[R][2]: index 0 = soft-demapper stage (relative to the aligned sequence), 1 =
 * constellation stage; workspace: vaeq_dp_epilogue_ws_bytes(R, N). */
int vaeq_cma_epilogue(int32_t R, int64_t N, int32_t n_lev, const float *y, const void *tx_f16, const float *amp, const float *var,
                      const float *nu_sc, float *ser, int32_t *shift, int32_t *rflag, void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * Seeded on-device DP channel simulator (input producer): optical_DP_channel/shared_funcs.py:65-90 in three stages with
 * the FFT / inverse FFT over Ls done by the caller (hipFFT through torch.fft) between them.  Counter-based RNG (Philox4x32-10):
 * every value is a function of (seed, frame, run, stream, index) only.
 *   vaeq_gen_dp_tx      : PCS draw (cdf[R][n_lev] = cumulative pmf), zero-stuffing, 'valid' FIR with g[Lg] = h_pulse * h_channel
 *                         (complex, interleaved) -> sig[R][2][Ls] complex64 (interleaved), Ls = sps*(N_conv-1)+1 - Lg + 1;
 *                         data_f16 (nullable) [R][2][2][N]: TX reference = symbols ref_offset .. ref_offset+N-1 (:89)
 *   vaeq_gen_dp_disperse: spectrum x H(f) (PMD + rotation theta[r] + IQ phase e_k = exp(-j phiIQ[k])) x CD phase (:38-54), in place;
 *                         fs = symb_rate * sps; scale multiplies the result (1/Ls folds the inverse FFT's normalisation in)
 *   vaeq_gen_dp_finish  : sigma_n from the mean power (:83), complex AWGN (:84), planar rx[R][2][2][sps*N] (:88); power_ws[R] scratch,
 *                         sigma_out[R] nullable
 * Rows of sig are Lrow >= Ls complex samples long; stage 1 zero-fills [Ls, Lrow).  Lrow == Ls reproduces the reference's circular
 * filtering over the exact sequence length; a padded Lrow (a fast FFT length) turns it into linear filtering -- only the few samples
 * within the dispersion's impulse-response length of the frame edges differ. */
int vaeq_gen_dp_tx(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t Lrow,
                   int32_t ref_offset, const float *amp, const float *cdf, const float *g_complex, uint64_t seed, uint32_t frame,
                   float *sig_complex, void *data_f16, void *stream);
int vaeq_gen_dp_disperse(int32_t R, int32_t Ls, double fs, double tau_cd, double tau_pmd, float e0_re, float e0_im, float e1_re,
                         float e1_im, float scale, const float *theta, float *spec_complex, void *stream);
int vaeq_gen_dp_finish(int32_t R, int32_t N, int32_t sps, int32_t Ls, int32_t Lrow, const float *snr_db, uint64_t seed, uint32_t frame,
                       const float *sig_complex, float *power_ws, float *rx, float *sigma_out, void *stream);

/* One DP frame in ONE call; same arguments as the stage entry points, e_k = exp(-j phiIQ[k]), fs = symb_rate * sps.  sig_ws[R][2][Lrow]
 * complex64 is scratch.  Two implementations of the same model, same Philox words (results agree to transform rounding):
 *   fused   sps == 2 and Lrow = N1 * 1024 with N1 in {4, 5, 8, 10, 16, 20} (the default frame pads to 20 * 1024): three passes over the signal --
 *           pulse shaping + the N1-point outer DFT stage in registers; per (run, k1) one wavefront: 1024-point FFT, fibre matrix, inverse FFT
 *           (rows through LDS, in place); inverse outer stage + noise + planar split.  No hipFFT.  Twiddles and the per-frequency phases of the
 *           fibre live in library-owned device tables, built once per (device, Lrow, fs, tau_cd, tau_pmd) and immutable afterwards.
 *   staged  any other shape (or env VAEQ_GEN_STAGED=1): the three stage kernels around in-place hipFFT transforms, plans cached per (Lrow, R).
 * power_ws: [R][vaeq_gen_dp_power_parts(Lrow)] floats -- for sps == 2 the first pass leaves partial sums of |sig|^2 there and the noise level
 * (:83) is derived from them: the fibre's transfer matrix is unitary at every frequency (:38-54), the dispersed signal has the power of the
 * undispersed one, so no pass over the dispersed signal is spent on it (other sps: the first R floats, filled by a power pass as in
 * vaeq_gen_dp_finish). */
int32_t vaeq_gen_dp_power_parts(int32_t Lrow);               /* max(8, 2 * ceil(Lrow / 2048)) */
int vaeq_gen_dp_frame(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t Lrow,
                      int32_t ref_offset, const float *amp, const float *cdf, const float *g_complex, const float *snr_db,
                      const float *theta, double fs, double tau_cd, double tau_pmd, float e0_re, float e0_im, float e1_re, float e1_im,
                      uint64_t seed, uint32_t frame, float *sig_ws, float *power_ws, float *rx, void *data_f16, float *sigma_out,
                      void *stream);

/* ------------------------------------------------------------------------
 * the AWGN VAE-NN equalizer (AWGN_channel/func_VAENN_MQAM.py).  Net (:170-188) = Conv1d(2, C, k1, pad k1/2) -> ELU ->
 * Conv1d(C, C, k2, pad k2/2, stride sps) -> per-axis softmax, C = 2 n_lev; loss_function (:63-95); Adam(amsgrad=True) on all
 * parameters (:248-253).  One run's parameters are ONE flat vector in the order of net.parameters() followed by h_est:
 *   theta = [fc1.weight C*2*k1 | fc1.bias C | fc2.weight C*C*k2 | fc2.bias C | h_est 2*M],  vaeq_nn_param_count() floats;
 * the Adam vectors (m, v, max v) and dbg_g use the same layout.  Net_BN (:190-211, batch_norm = 1): BatchNorm1d(C) between the ELU
 * and fc2; theta gains [batch1.weight C | batch1.bias C] in front of h_est, and bn_running[R][2][C] = (running_mean, running_var)
 * is caller-owned state updated by every training step (momentum 0.1) and read by the eval-mode entry points.
 * vaeq_nn_train replaces the minibatch loop (:274-285) for R runs: step s uses symbols [s*B, (s+1)*B) of rx[R][2][S]. */
typedef struct vaeq_nn_args {
    int32_t R, steps, B, sps, M, n_lev, k1, k2;
    int64_t S;
    const float *rx;     /* [R][2][S] */
    float *theta;        /* [R][NP] in/out */
    float *adam_m;       /* [R][NP] in/out */
    float *adam_v;       /* [R][NP] in/out */
    float *adam_x;       /* [R][NP] in/out: max_exp_avg_sq */
    int32_t *step;       /* [R] in/out */
    const float *amp;    /* [n_lev] */
    const float *lr;     /* [R] */
    float *loss;         /* nullable [R][steps] */
    float *q_out;        /* nullable [R][2*n_lev][steps*B] */
    float *dbg_g;        /* nullable [R][NP]: gradient of the LAST step */
    int32_t no_update;   /* 1: skip the Adam update (and the running-statistics update) */
    int32_t batch_norm;  /* 0: Net, 1: Net_BN */
    float *bn_running;   /* Net_BN: [R][2][C] in/out */
} vaeq_nn_args;

int vaeq_nn_train(const vaeq_nn_args *args, void *stream);
int64_t vaeq_nn_param_count(int32_t M, int32_t n_lev, int32_t k1, int32_t k2, int32_t batch_norm);
int64_t vaeq_nn_lds_bytes(int32_t B, int32_t sps, int32_t M, int32_t n_lev, int32_t k1, int32_t k2, int32_t batch_norm);
/* Net.forward in eval mode on N symbols per run (:293-295): x[R][2][N*sps], theta[R][NP] -> q[R][2*n_lev][N].
 * bn_running: NULL for Net; [R][2][C] for Net_BN (net.eval(): the running statistics normalise). */
int vaeq_nn_forward(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t k1, int32_t k2, const float *x,
                    const float *theta, const float *bn_running, float *q, void *stream);

/* The whole VAE-NN validation block (:287-301: eval forward, find_shift :147-166, SER_q :97-123) in one call, q not materialised:
 * x[R][2][N*sps], theta[R][NP], data_f16[R][2][N] -> ser[R], shift[R] (nullable). */
int vaeq_nn_validate(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t k1, int32_t k2, int32_t n_shift, const float *x,
                     const float *theta, const float *bn_running, const float *amp, const void *data_f16, float *ser, int32_t *shift,
                     void *stream);

/* Single-polarisation AWGN / ISI channel of AWGN_channel/func_VAELE_MQAM_shaping.py:39-61 (generate_data) for R runs, same three
 * stages without the dispersion step: g[Lg] = rrc * h_channel; scratch: power_ws [R][ceil(Ls / 2048)] floats, and sig_ws [R][Ls]
 * complex64 only for sps != 2 (for sps == 2 the clean signal stays in registers: frames of up to four 2048-sample tiles in ONE pass, one workgroup per
 * run; longer ones in two -- one pass for the power, one that adds the noise -- with bit-identical results; env VAEQ_AWGN_TWOPASS=1 forces two);
 * rx[R][2][sps*N] (:57), data_f16 (nullable) [R][2][N] = symbols ref_offset .. ref_offset+N-1 (:59), sigma_out[R] nullable.
 * sigma_fixed (nullable [R]): use this noise standard deviation instead of the power-derived one -- the VAE-NN script's model
 * (func_VAENN_MQAM.py:52: sigma_n = sqrt(1/2) / 10^(SNR/20)); snr_db may then be NULL. */
int vaeq_gen_awgn(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t ref_offset,
                  const float *amp, const float *cdf, const float *g_complex, const float *snr_db, uint64_t seed, uint32_t frame,
                  float *sig_ws, float *power_ws, float *rx, void *data_f16, float *sigma_out, const float *sigma_fixed, void *stream);

/* Fused validation pass of one AWGN epoch (func_VAELE_MQAM_shaping.py:308-318): twoFIR.forward in eval mode on N symbols per run,
 * find_shift (:188-204, n_shift circular lags over the first 1000 symbols) and SER_q (:97-123, argmax decisions, minimum over the
 * four quadrant rotations, 11 symbols trimmed at both ends) without materialising q.
 * x[R][2][N*sps], W[R][2][M], data_f16[R][2][N] -> ser[R], shift[R] (nullable); y_ws[R][2][N] receives the un-normalised output. */
int vaeq_awgn_validate(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t n_shift, const float *x, const float *W,
                       const float *amp, const float *amp_mean, const float *var, const void *data_f16, float *y_ws, float *ser,
                       int32_t *shift, void *stream);

/* The validation frame of an epoch without its round trip through HBM (func_VAELE_MQAM_shaping.py:310 generate_data followed by :311-318): the
 * reference draws N_valid = 15 000 fresh symbols per evaluated epoch, 12.5 x what it trains on, and reads them once.
 * vaeq_gen_awgn_clean = vaeq_gen_awgn's first stage alone (sps == 2 only): the noise-free channel output sig[R][Ls] (complex64), the power sums
 * power_ws[R][ceil(Ls / 2048)] of its tiles and the TX reference data_f16[R][2][N].
 * vaeq_awgn_validate_gen = vaeq_awgn_validate on x = sig + noise, the noise added where a tile is staged: the Philox words, sigma_n (from
 * power_ws and snr_db, or sigma_fixed) and the arithmetic of vaeq_gen_awgn -- ser / shift / y_ws are bit for bit what vaeq_gen_awgn followed by
 * vaeq_awgn_validate return for the same (seed, frame) (sps == 2, M in {9, 17, 25}; VAEQ_ERR_SHAPE otherwise); sigma_out[R] nullable. */
int vaeq_gen_awgn_clean(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t ref_offset,
                        const float *amp, const float *cdf, const float *g_complex, uint64_t seed, uint32_t frame, float *sig,
                        float *power_ws, void *data_f16, void *stream);
int vaeq_awgn_validate_gen(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t n_shift, const float *sig, int32_t Ls,
                           const float *power_ws, const float *snr_db, const float *sigma_fixed, uint64_t seed, uint32_t frame,
                           const float *W, const float *amp, const float *amp_mean, const float *var, const void *data_f16, float *y_ws,
                           float *ser, int32_t *shift, float *sigma_out, void *stream);

/* ------------------------------------------------------------------------
 * the constant-modulus baselines of the DP scripts and their carrier phase estimation.
 * vaeq_cma: CMA (shared_funcs.py:341-383, mode 0) / CMAbatch (:385-433, mode 1 with symb_step = batchlen) / CMAflex (:435-488,
 * mode 1) on one frame per run: rx[R][2][2][N] -> out[R][2][2][N/sps], e[R][N/sps][2] (nullable); taps h[R][2][2][2][M] and the
 * per-run step size lr[R] as in the reference (h is updated in place; R_mod is the modulus constant `R` of the reference).
 * vaeq_cpe: Viterbi-Viterbi carrier phase estimation (:139-186), y[R][2][2][N] -> y_out[R][2][2][N], M_ma = 501 in the reference. */
int vaeq_cma(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t mode, int32_t batchlen, int32_t symb_step, const float *rx, float R_mod,
             float *h, const float *lr, float *out, float *e, void *stream);
int vaeq_cpe(int32_t R, int64_t N, int32_t M_ma, const float *y, float *y_out, void *stream);

int vaeq_version(void);
const char *vaeq_strerror(int code);

/* Measurement helpers (no reference counterpart; SURVEY 8d asks for them).
 * vaeq_last_kernel: name of the kernel instantiation the calling thread's most recent vaeq_dp_train / vaeq_awgn_train launched (as a profiler
 * prints it, e.g. "vaeq::dp_wave_kernel<25, 8, 100, true, 1, 1>"), copied into buf[len] -- bench.py names its roofline kernel from this.
 * vaeq_stream_copy: dst[bytes] = src[bytes] with a plain 16-byte grid-stride copy kernel (bytes and both pointers multiples of 16): the
 * measured HBM copy bandwidth that stands next to the 8 TB/s spec peak in the roofline. */
int vaeq_last_kernel(char *buf, int32_t len);
int vaeq_stream_copy(void *dst, const void *src, int64_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VAEQ_H */
