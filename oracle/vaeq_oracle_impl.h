/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C) of the VAE blind-equalizer training inner loop of
 * kit-cel/vae-equalizer.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (vae_equalizer_amd) never
 * imports, links or calls it.
 *
 * This header is the implementation body.  vaeq_oracle.c includes it twice, once
 * with REAL=float (suffix _f32: the reference's own arithmetic type) and once
 * with REAL=double (suffix _f64: the "truth" both the reference's goldens and
 * the HIP kernels are measured against).
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function below
 * against vectors captured from the reference itself (tools/capture_golden.py,
 * fixtures under tests/golden/).
 *
 * All file:line citations are relative to the reference tree:
 *   DP   = optical_DP_channel/shared_funcs.py
 *   LEDP = optical_DP_channel/func_VAELE_DP_MQAM_shaping.py
 *   FLEX = optical_DP_channel/func_VAEflex_DP_MQAM_shaping.py
 *   AWGN = AWGN_channel/func_VAELE_MQAM_shaping.py
 *
 * The backward pass is written the way autograd walks the reference's forward
 * graph (explicit zero-stuffed Eq/Var grids, per-level dL/dq, softmin backward,
 * channel-packed conv weight gradient).  The HIP kernels use an independently
 * simplified closed form (central moments), so the two derivations check each
 * other.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#define MAXLEV 16

/* ---------------------------------------------------------------- helpers */
static inline REAL FN(r_exp)(REAL x) { return (REAL)exp((double)x); }
static inline REAL FN(r_log)(REAL x) { return (REAL)log((double)x); }
static inline REAL FN(r_sqrt)(REAL x) { return (REAL)sqrt((double)x); }
static inline REAL FN(r_abs)(REAL x) { return x < 0 ? -x : x; }

/* softmin over n levels of cost c_i (nn.Softmin(dim=0), DP:497 / AWGN:212): softmax(-c), max-subtracted */
static void FN(softmin)(int n, const REAL *cost, REAL *q)
{
    REAL cmin = cost[0], s = 0;
    for (int i = 1; i < n; i++) if (cost[i] < cmin) cmin = cost[i];
    for (int i = 0; i < n; i++) { q[i] = FN(r_exp)(-(cost[i] - cmin)); s += q[i]; }
    for (int i = 0; i < n; i++) q[i] = q[i] / s;
}

/* ------------------------------------------------------------------ R1+R2
 * twoXtwoFIR.forward, DP:500-527.  x[2][2][L] (pol, I/Q, sample), W[2][4][M]
 * (Conv1d(4->2,k=M,stride=sps,pad=M//2), DP:494), channel packing DP:505,507:
 *   x_in_I = [xI0, xI1, -xQ0, -xQ1],  x_in_Q = [xQ0, xQ1, xI0, xI1].
 * q[2][2n][B] rows 0..n-1 = I levels, n..2n-1 = Q levels; out[2][2][B]. */
void FN(vaeq_oracle_dp_forward)(int B, int sps, int M, int n, const REAL *x, const REAL *W, const REAL *amp,
                                const REAL *var, REAL nu_sc, REAL *q, REAL *out)
{
    const int L = B * sps, pad = M / 2;
    for (int o = 0; o < 2; o++)
        for (int nn = 0; nn < B; nn++) {
            REAL accI = 0, accQ = 0;
            for (int ch = 0; ch < 4; ch++) {
                const int p = ch & 1, neg = ch >> 1;
                for (int k = 0; k < M; k++) {
                    const int s = nn * sps + k - pad;
                    if (s < 0 || s >= L) continue;            /* zero padding, DP:494 */
                    const REAL xi = x[(p * 2 + 0) * L + s], xq = x[(p * 2 + 1) * L + s];
                    const REAL inI = neg ? -xq : xi;          /* DP:505 */
                    const REAL inQ = neg ? xi : xq;           /* DP:507 */
                    const REAL w = W[(o * 4 + ch) * M + k];
                    accI += w * inI;
                    accQ += w * inQ;
                }
            }
            out[(o * 2 + 0) * B + nn] = accI;                 /* DP:518 */
            out[(o * 2 + 1) * B + nn] = accQ;
            for (int c = 0; c < 2; c++) {                     /* DP:521-523 */
                const REAL y = c ? accQ : accI;
                REAL cost[MAXLEV], qq[MAXLEV];
                for (int i = 0; i < n; i++) {
                    const REAL d = y - amp[i];
                    cost[i] = d * d / 2 / var[o] + nu_sc * (amp[i] * amp[i]);
                }
                FN(softmin)(n, cost, qq);
                for (int i = 0; i < n; i++) q[(o * 2 * n + c * n + i) * B + nn] = qq[i];
            }
        }
}

/* soft_dec, DP:529-542: the demapper alone on out[2][2][N] */
void FN(vaeq_oracle_dp_soft_dec)(int N, int n, const REAL *out, const REAL *var, const REAL *amp, REAL nu_sc, REAL *q)
{
    for (int o = 0; o < 2; o++)
        for (int c = 0; c < 2; c++)
            for (int nn = 0; nn < N; nn++) {
                const REAL y = out[(o * 2 + c) * N + nn];
                REAL cost[MAXLEV], qq[MAXLEV];
                for (int i = 0; i < n; i++) {
                    const REAL d = y - amp[i];
                    cost[i] = d * d / 2 / var[o] + nu_sc * (amp[i] * amp[i]);
                }
                FN(softmin)(n, cost, qq);
                for (int i = 0; i < n; i++) q[(o * 2 * n + c * n + i) * N + nn] = qq[i];
            }
}

/* --------------------------------------------------------------------- R3
 * loss_function_shaping, DP:92-137.  q[2][2n][B], x[2][2][L], h[2][2][2][M]
 * (chi, nu, re/im, tap).  Returns loss; var_est[2] = C/(N-Mh) (DP:137).
 * Optional work arrays (may be NULL) expose the intermediates the backward needs. */
static REAL FN(dp_loss_core)(int B, int sps, int M, int n, const REAL *q, const REAL *x, const REAL *h, const REAL *amp,
                             const REAL *P, REAL *var_est, REAL *Eq /*[2][2][L]*/, REAL *Var /*[2][2][L]*/,
                             REAL *Dre /*[2][nm]*/, REAL *Dim, REAL *C /*[2]*/)
{
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh;
    for (int i = 0; i < 2 * 2 * L; i++) { Eq[i] = 0; Var[i] = 0; }
    /* DP:107-113: E_q[x], E_q[x^2] on the zero-stuffed grid, Var = E[x^2]-E[x]^2 */
    for (int v = 0; v < 2; v++)
        for (int c = 0; c < 2; c++)
            for (int nn = 0; nn < B; nn++) {
                REAL mu = 0, rho = 0;
                for (int i = 0; i < n; i++) {
                    const REAL qq = q[(v * 2 * n + c * n + i) * B + nn];
                    mu += amp[i] * qq;
                    rho += (amp[i] * amp[i]) * qq;
                }
                Eq[(v * 2 + c) * L + nn * sps] = mu;
                Var[(v * 2 + c) * L + nn * sps] = rho - mu * mu;
            }
    REAL E[2] = {0, 0};
    for (int i = 0; i < 2 * nm; i++) { Dre[i] = 0; Dim[i] = 0; }
    /* DP:123-129 */
    for (int j = 0; j <= Mh; j++) {
        REAL vs[2] = {0, 0};
        for (int v = 0; v < 2; v++)
            for (int c = 0; c < 2; c++)
                for (int t = 0; t < nm; t++) vs[v] += Var[(v * 2 + c) * L + t + Mh - j];   /* DP:128 */
        for (int chi = 0; chi < 2; chi++) {
            for (int v = 0; v < 2; v++) {
                const REAL hr = h[((chi * 2 + v) * 2 + 0) * M + j], hi = h[((chi * 2 + v) * 2 + 1) * M + j];
                for (int t = 0; t < nm; t++) {
                    const REAL eI = Eq[(v * 2 + 0) * L + t + Mh - j], eQ = Eq[(v * 2 + 1) * L + t + Mh - j];
                    Dre[chi * nm + t] += hr * eI - hi * eQ;    /* DP:124-125 */
                    Dim[chi * nm + t] += hi * eI + hr * eQ;    /* DP:126-127 */
                }
                E[chi] += (hr * hr + hi * hi) * vs[v];         /* DP:115,129 */
            }
        }
    }
    /* DP:131-132: q log(q/P + 1e-12) over SYMBOL indices mh .. B-mh-1 */
    REAL ent = 0;
    for (int v = 0; v < 2; v++)
        for (int r = 0; r < 2 * n; r++)
            for (int nn = mh; nn < B - mh; nn++) {
                const REAL qq = q[(v * 2 * n + r) * B + nn];
                ent += -qq * FN(r_log)(qq / P[r % n] + (REAL)1e-12);
            }
    /* DP:133-134 */
    REAL loss = 0;
    for (int chi = 0; chi < 2; chi++) {
        REAL s_xx = 0, s_xd = 0, s_dd = 0;
        for (int t = 0; t < nm; t++) {
            const REAL xr = x[(chi * 2 + 0) * L + mh + t], xi = x[(chi * 2 + 1) * L + mh + t];
            s_xx += xr * xr + xi * xi;
            s_xd += xr * Dre[chi * nm + t] + xi * Dim[chi * nm + t];
            s_dd += Dre[chi * nm + t] * Dre[chi * nm + t] + Dim[chi * nm + t] * Dim[chi * nm + t];
        }
        C[chi] = s_xx - 2 * s_xd + s_dd + E[chi];
        loss += (REAL)nm * FN(r_log)(C[chi]);                   /* DP:136 */
        var_est[chi] = C[chi] / (REAL)nm;                       /* DP:137 */
    }
    return loss - ent;
}

REAL FN(vaeq_oracle_dp_loss)(int B, int sps, int M, int n, const REAL *q, const REAL *x, const REAL *h, const REAL *amp,
                             const REAL *P, REAL *var_est)
{
    const int L = B * sps, nm = L - 2 * (M / 2);
    REAL *buf = (REAL *)malloc(sizeof(REAL) * (8 * L + 4 * nm));
    REAL C[2];
    REAL loss = FN(dp_loss_core)(B, sps, M, n, q, x, h, amp, P, var_est, buf, buf + 4 * L, buf + 8 * L, buf + 8 * L + 2 * nm, C);
    free(buf);
    return loss;
}

/* --------------------------------------------------------------------- R4
 * loss.backward() (LEDP:65) restated: forward DP:500-527 + DP:92-137, then the
 * chain rule node by node.  Outputs q, out, loss, var_est and gW[2][4][M],
 * gh[2][2][2][M]. */
REAL FN(vaeq_oracle_dp_step_grads)(int B, int sps, int M, int n, const REAL *x, const REAL *W, const REAL *h,
                                   const REAL *amp, const REAL *P, const REAL *var, REAL nu_sc, REAL *q, REAL *out,
                                   REAL *var_est, REAL *gW, REAL *gh)
{
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh, pad = M / 2;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * (16 * L + 4 * nm + 4 * B));
    REAL *Eq = buf, *Var = buf + 4 * L, *gEq = buf + 8 * L, *gVar = buf + 12 * L;
    REAL *Dre = buf + 16 * L, *Dim = Dre + 2 * nm, *gy = Dim + 2 * nm; /* gy[2][2][B] */
    REAL C[2], gC[2];

    FN(vaeq_oracle_dp_forward)(B, sps, M, n, x, W, amp, var, nu_sc, q, out);
    REAL loss = FN(dp_loss_core)(B, sps, M, n, q, x, h, amp, P, var_est, Eq, Var, Dre, Dim, C);

    for (int chi = 0; chi < 2; chi++) gC[chi] = (REAL)nm / C[chi];          /* d(nm log C)/dC */
    for (int i = 0; i < 4 * L; i++) { gEq[i] = 0; gVar[i] = 0; }
    for (int i = 0; i < 2 * 2 * 2 * M; i++) gh[i] = 0;

    /* dC/dD = -2 (x - D)   (from  sum x^2 - 2 sum x.D + sum D^2, DP:133-134) */
    for (int chi = 0; chi < 2; chi++)
        for (int v = 0; v < 2; v++)
            for (int j = 0; j <= Mh; j++) {
                const REAL hr = h[((chi * 2 + v) * 2 + 0) * M + j], hi = h[((chi * 2 + v) * 2 + 1) * M + j];
                REAL ghr = 0, ghi = 0, vs = 0;
                for (int t = 0; t < nm; t++) {
                    const int s = t + Mh - j;
                    const REAL dDr = -2 * (x[(chi * 2 + 0) * L + mh + t] - Dre[chi * nm + t]) * gC[chi];
                    const REAL dDi = -2 * (x[(chi * 2 + 1) * L + mh + t] - Dim[chi * nm + t]) * gC[chi];
                    const REAL eI = Eq[(v * 2 + 0) * L + s], eQ = Eq[(v * 2 + 1) * L + s];
                    /* D_re += hr*eI - hi*eQ ; D_im += hi*eI + hr*eQ   (DP:124-127) */
                    ghr += dDr * eI + dDi * eQ;
                    ghi += -dDr * eQ + dDi * eI;
                    gEq[(v * 2 + 0) * L + s] += dDr * hr + dDi * hi;
                    gEq[(v * 2 + 1) * L + s] += -dDr * hi + dDi * hr;
                    /* E += |h|^2 * Var_sum  (DP:128-129) */
                    gVar[(v * 2 + 0) * L + s] += gC[chi] * (hr * hr + hi * hi);
                    gVar[(v * 2 + 1) * L + s] += gC[chi] * (hr * hr + hi * hi);
                    vs += Var[(v * 2 + 0) * L + s] + Var[(v * 2 + 1) * L + s];
                }
                gh[((chi * 2 + v) * 2 + 0) * M + j] = ghr + gC[chi] * 2 * hr * vs;
                gh[((chi * 2 + v) * 2 + 1) * M + j] = ghi + gC[chi] * 2 * hi * vs;
            }
    /* Var = Eq2 - Eq^2 (DP:113) */
    for (int i = 0; i < 4 * L; i++) gEq[i] += -2 * Eq[i] * gVar[i];

    /* per symbol: dL/dq_i -> softmin backward -> dL/dy  (DP:107-112, 131-132, 521-523) */
    for (int o = 0; o < 2; o++)
        for (int c = 0; c < 2; c++)
            for (int nn = 0; nn < B; nn++) {
                const REAL gmu = gEq[(o * 2 + c) * L + nn * sps], grho = gVar[(o * 2 + c) * L + nn * sps];
                const REAL y = out[(o * 2 + c) * B + nn];
                const int inr = (nn >= mh && nn < B - mh);
                REAL gq[MAXLEV], dot = 0;
                for (int i = 0; i < n; i++) {
                    const REAL qq = q[(o * 2 * n + c * n + i) * B + nn];
                    gq[i] = amp[i] * gmu + (amp[i] * amp[i]) * grho;
                    if (inr) {
                        const REAL r = qq / P[i], re = r + (REAL)1e-12;
                        gq[i] += FN(r_log)(re) + r / re;        /* d/dq [ q log(q/P+eps) ] */
                    }
                    dot += qq * gq[i];
                }
                REAL g = 0;
                for (int i = 0; i < n; i++) {
                    const REAL qq = q[(o * 2 * n + c * n + i) * B + nn];
                    const REAL gz = qq * (gq[i] - dot);          /* softmax backward wrt logits z=-cost */
                    g += gz * (-(y - amp[i]) / var[o]);           /* dz_i/dy */
                }
                gy[(o * 2 + c) * B + nn] = g;
            }
    /* conv weight gradient through the channel packing DP:505,507 */
    for (int o = 0; o < 2; o++)
        for (int ch = 0; ch < 4; ch++) {
            const int p = ch & 1, neg = ch >> 1;
            for (int k = 0; k < M; k++) {
                REAL acc = 0;
                for (int nn = 0; nn < B; nn++) {
                    const int s = nn * sps + k - pad;
                    if (s < 0 || s >= L) continue;
                    const REAL xi = x[(p * 2 + 0) * L + s], xq = x[(p * 2 + 1) * L + s];
                    const REAL inI = neg ? -xq : xi, inQ = neg ? xi : xq;
                    acc += gy[(o * 2 + 0) * B + nn] * inI + gy[(o * 2 + 1) * B + nn] * inQ;
                }
                gW[(o * 4 + ch) * M + k] = acc;
            }
        }
    free(buf);
    return loss;
}

/* --------------------------------------------------------------------- R5
 * torch.optim.Adam single-tensor step (LEDP:28,31,66; AWGN:283 with amsgrad).
 * Scalar prep in double like the Python side of torch.optim.adam; tensor math in REAL.
 * step is the step count AFTER the increment (1 for the first update). */
void FN(vaeq_oracle_adam)(int cnt, REAL *p, const REAL *g, REAL *m, REAL *v, REAL *vmax, int step, double lr, int amsgrad)
{
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const REAL step_size = (REAL)(lr / bc1), bc2_sqrt = (REAL)sqrt(bc2);
    const REAL w1 = (REAL)(1.0 - b1), w2 = (REAL)(1.0 - b2), fb2 = (REAL)b2, feps = (REAL)eps;
    for (int i = 0; i < cnt; i++) {
#if REAL_IS_FLOAT
        m[i] = fmaf(g[i] - m[i], w1, m[i]);                      /* exp_avg.lerp_(grad, 1-beta1) */
#else
        m[i] = fma(g[i] - m[i], w1, m[i]);
#endif
        v[i] = v[i] * fb2;
        v[i] = v[i] + w2 * g[i] * g[i];                          /* mul_(beta2).addcmul_(g,g,1-beta2) */
        REAL vv = v[i];
        if (amsgrad) { if (v[i] > vmax[i]) vmax[i] = v[i]; vv = vmax[i]; }
        const REAL denom = FN(r_sqrt)(vv) / bc2_sqrt + feps;
        p[i] = p[i] + (-step_size * m[i]) / denom;               /* addcdiv_(m, denom, -step_size) */
    }
}

/* ------------------------------------------------------------------ R6/R7
 * One frame of the DP minibatch loop for ONE run.
 *   VAE-LE (LEDP:57-66):  stride = keep_len = B, keep_off = 0
 *   VAEflex (FLEX:59-70): stride = keep_len = flex_step, keep_off = (B-flex_step)/2
 * rx[2][2][S]; window of step s starts at symbol s*stride.  q_out[2][2n][n_steps*keep_len],
 * y_out[2][2][n_steps*keep_len], loss[n_steps], var_est[2][n_steps].  State (W,h,Adam m/v,
 * step counter) is caller-owned and updated in place. */
void FN(vaeq_oracle_dp_train)(int n_steps, int B, int sps, int M, int n, int stride, int keep_off, int keep_len, int S,
                              const REAL *rx, REAL *W, REAL *h, REAL *mW, REAL *vW, REAL *mh_, REAL *vh, int *step,
                              const REAL *amp, const REAL *P, const REAL *var, REAL nu_sc, double lr_W, double lr_h,
                              REAL *q_out, REAL *y_out, REAL *loss, REAL *var_est)
{
    const int L = B * sps, No = n_steps * keep_len;
    REAL *mb = (REAL *)malloc(sizeof(REAL) * (4 * L + 4 * n * B + 4 * B + 8 * M + 8 * M));
    REAL *q = mb + 4 * L, *out = q + 4 * n * B, *gW = out + 4 * B, *gh = gW + 8 * M;
    for (int s = 0; s < n_steps; s++) {
        const int s0 = s * stride * sps;
        for (int r = 0; r < 4; r++) memcpy(mb + r * L, rx + (size_t)r * S + s0, sizeof(REAL) * L);   /* LEDP:58 / FLEX:60 */
        REAL ve[2];
        loss[s] = FN(vaeq_oracle_dp_step_grads)(B, sps, M, n, mb, W, h, amp, P, var, nu_sc, q, out, ve, gW, gh);
        var_est[0 * n_steps + s] = ve[0];
        var_est[1 * n_steps + s] = ve[1];
        if (q_out)
            for (int r = 0; r < 4 * n; r++)
                memcpy(q_out + (size_t)r * No + s * keep_len, q + r * B + keep_off, sizeof(REAL) * keep_len);
        if (y_out)
            for (int r = 0; r < 4; r++)
                memcpy(y_out + (size_t)r * No + s * keep_len, out + r * B + keep_off, sizeof(REAL) * keep_len);
        *step += 1;
        FN(vaeq_oracle_adam)(8 * M, W, gW, mW, vW, NULL, *step, lr_W, 0);
        FN(vaeq_oracle_adam)(8 * M, h, gh, mh_, vh, NULL, *step, lr_h, 0);
    }
    free(mb);
}

/* ================================================================== AWGN */

/* R8: twoFIR.forward, AWGN:214-231.  x[2][L], W[1][2][M] (pad=(M-1)/2, AWGN:209):
 *   out_I = W0*x0 + W1*x1,  out_Q = W0*x1 - W1*x0   (AWGN:216-219)
 * normalised yhat_c = y_c / mean|y_c| * amp_mean (AWGN:228); q = softmin((yhat-a)^2/var) (AWGN:229).
 * Returns the UN-normalised out[2][B] (AWGN:227,231) and q[2n][B]; ynorm (may be NULL) gets yhat. */
void FN(vaeq_oracle_awgn_forward)(int B, int sps, int M, int n, const REAL *x, const REAL *W, const REAL *amp,
                                  REAL amp_mean, REAL var, REAL *q, REAL *out, REAL *ynorm, REAL *mabs)
{
    const int L = B * sps, pad = (M - 1) / 2;
    for (int nn = 0; nn < B; nn++) {
        REAL aI = 0, aQ = 0;
        for (int k = 0; k < M; k++) {
            const int s = nn * sps + k - pad;
            if (s < 0 || s >= L) continue;
            aI += W[k] * x[s] + W[M + k] * x[L + s];
            aQ += W[k] * x[L + s] - W[M + k] * x[s];
        }
        out[nn] = aI;
        out[B + nn] = aQ;
    }
    for (int c = 0; c < 2; c++) {
        REAL ma = 0;
        for (int nn = 0; nn < B; nn++) ma += FN(r_abs)(out[c * B + nn]);
        ma /= (REAL)B;
        if (mabs) mabs[c] = ma;
        for (int nn = 0; nn < B; nn++) {
            const REAL yh = out[c * B + nn] / ma * amp_mean;
            if (ynorm) ynorm[c * B + nn] = yh;
            REAL cost[MAXLEV], qq[MAXLEV];
            for (int i = 0; i < n; i++) { const REAL d = yh - amp[i]; cost[i] = d * d / var; }
            FN(softmin)(n, cost, qq);
            for (int i = 0; i < n; i++) q[(c * n + i) * B + nn] = qq[i];
        }
    }
}

/* R9: loss_function, AWGN:63-95.  h[2][M] (re, im). */
static REAL FN(awgn_loss_core)(int B, int sps, int M, int n, const REAL *q, const REAL *x, const REAL *h, const REAL *amp,
                               const REAL *P, REAL *Eq /*[2][L]*/, REAL *Eq2, REAL *Dre /*[nm]*/, REAL *Dim, REAL *Cout)
{
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh;
    for (int i = 0; i < 2 * L; i++) { Eq[i] = 0; Eq2[i] = 0; }
    for (int c = 0; c < 2; c++)
        for (int nn = 0; nn < B; nn++) {
            REAL mu = 0, rho = 0;
            for (int i = 0; i < n; i++) {
                const REAL qq = q[(c * n + i) * B + nn];
                mu += amp[i] * qq;
                rho += (amp[i] * amp[i]) * qq;
            }
            Eq[c * L + nn * sps] = mu;                           /* AWGN:77 */
            Eq2[c * L + nn * sps] = rho;                         /* AWGN:78 */
        }
    REAL sE = 0;
    for (int t = 0; t < nm; t++) { Dre[t] = 0; Dim[t] = 0; }
    for (int j = 0; j <= Mh; j++) {                              /* AWGN:85-88 */
        const REAL hr = h[j], hi = h[M + j];
        for (int t = 0; t < nm; t++) {
            const int s = t + Mh - j;
            Dre[t] += hr * Eq[s] - hi * Eq[L + s];
            Dim[t] += hr * Eq[L + s] + hi * Eq[s];
            sE += (hr * hr + hi * hi) * ((Eq2[s] - Eq[s] * Eq[s]) + (Eq2[L + s] - Eq[L + s] * Eq[L + s]));
        }
    }
    REAL ent = 0;                                                /* AWGN:90-91 */
    for (int r = 0; r < 2 * n; r++)
        for (int nn = mh; nn < B - mh; nn++) {
            const REAL qq = q[r * B + nn];
            ent += -qq * FN(r_log)(qq / P[r % n] + (REAL)1e-12);
        }
    REAL s_xx = 0, s_xd = 0, s_dd = 0;                           /* AWGN:92-93 */
    for (int t = 0; t < nm; t++) {
        const REAL xr = x[mh + t], xi = x[L + mh + t];
        s_xx += xr * xr + xi * xi;
        s_xd += xr * Dre[t] + xi * Dim[t];
        s_dd += Dre[t] * Dre[t] + Dim[t] * Dim[t];
    }
    const REAL C = s_xx - 2 * s_xd + s_dd + sE;
    *Cout = C;
    return (REAL)nm * FN(r_log)(C) - ent;                        /* AWGN:94 */
}

REAL FN(vaeq_oracle_awgn_loss)(int B, int sps, int M, int n, const REAL *q, const REAL *x, const REAL *h, const REAL *amp,
                               const REAL *P)
{
    const int L = B * sps, nm = L - 2 * (M / 2);
    REAL *buf = (REAL *)malloc(sizeof(REAL) * (4 * L + 2 * nm));
    REAL C;
    REAL loss = FN(awgn_loss_core)(B, sps, M, n, q, x, h, amp, P, buf, buf + 2 * L, buf + 4 * L, buf + 4 * L + nm, &C);
    free(buf);
    return loss;
}

/* AWGN forward + loss + backward (AWGN:302-305): gW[1][2][M], gh[2][M] */
REAL FN(vaeq_oracle_awgn_step_grads)(int B, int sps, int M, int n, const REAL *x, const REAL *W, const REAL *h,
                                     const REAL *amp, const REAL *P, REAL amp_mean, REAL var, REAL *q, REAL *out,
                                     REAL *gW, REAL *gh)
{
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh, pad = (M - 1) / 2;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * (8 * L + 2 * nm + 6 * B));
    REAL *Eq = buf, *Eq2 = buf + 2 * L, *gEq = buf + 4 * L, *gEq2 = buf + 6 * L;
    REAL *Dre = buf + 8 * L, *Dim = Dre + nm, *yh = Dim + nm, *gyh = yh + 2 * B, *gy = gyh + 2 * B;
    REAL mabs[2], C;

    FN(vaeq_oracle_awgn_forward)(B, sps, M, n, x, W, amp, amp_mean, var, q, out, yh, mabs);
    REAL loss = FN(awgn_loss_core)(B, sps, M, n, q, x, h, amp, P, Eq, Eq2, Dre, Dim, &C);
    const REAL gC = (REAL)nm / C;
    for (int i = 0; i < 2 * L; i++) { gEq[i] = 0; gEq2[i] = 0; }
    for (int j = 0; j <= Mh; j++) {
        const REAL hr = h[j], hi = h[M + j], hh = hr * hr + hi * hi;
        REAL ghr = 0, ghi = 0, vs = 0;
        for (int t = 0; t < nm; t++) {
            const int s = t + Mh - j;
            const REAL dDr = -2 * (x[mh + t] - Dre[t]) * gC, dDi = -2 * (x[L + mh + t] - Dim[t]) * gC;
            /* D_re += hr*Eq0 - hi*Eq1 ; D_im += hr*Eq1 + hi*Eq0  (AWGN:86-87) */
            ghr += dDr * Eq[s] + dDi * Eq[L + s];
            ghi += -dDr * Eq[L + s] + dDi * Eq[s];
            gEq[s] += dDr * hr + dDi * hi;
            gEq[L + s] += -dDr * hi + dDi * hr;
            /* E term AWGN:88 */
            gEq2[s] += gC * hh;
            gEq2[L + s] += gC * hh;
            gEq[s] += gC * hh * (-2 * Eq[s]);
            gEq[L + s] += gC * hh * (-2 * Eq[L + s]);
            vs += (Eq2[s] - Eq[s] * Eq[s]) + (Eq2[L + s] - Eq[L + s] * Eq[L + s]);
        }
        gh[j] = ghr + gC * 2 * hr * vs;
        gh[M + j] = ghi + gC * 2 * hi * vs;
    }
    for (int c = 0; c < 2; c++)
        for (int nn = 0; nn < B; nn++) {
            const REAL gmu = gEq[c * L + nn * sps], grho = gEq2[c * L + nn * sps];
            const REAL y = yh[c * B + nn];
            const int inr = (nn >= mh && nn < B - mh);
            REAL gq[MAXLEV], dot = 0;
            for (int i = 0; i < n; i++) {
                const REAL qq = q[(c * n + i) * B + nn];
                gq[i] = amp[i] * gmu + (amp[i] * amp[i]) * grho;
                if (inr) {
                    const REAL r = qq / P[i], re = r + (REAL)1e-12;
                    gq[i] += FN(r_log)(re) + r / re;
                }
                dot += qq * gq[i];
            }
            REAL g = 0;
            for (int i = 0; i < n; i++) {
                const REAL qq = q[(c * n + i) * B + nn];
                g += qq * (gq[i] - dot) * (-2 * (y - amp[i]) / var);   /* cost=(yhat-a)^2/var, AWGN:229 */
            }
            gyh[c * B + nn] = g;
        }
    /* normalisation yhat = y / mean|y| * A  (AWGN:228) */
    for (int c = 0; c < 2; c++) {
        REAL dot = 0;
        for (int nn = 0; nn < B; nn++) dot += gyh[c * B + nn] * out[c * B + nn];
        for (int nn = 0; nn < B; nn++) {
            const REAL y = out[c * B + nn];
            const REAL sg = (y > 0) - (y < 0);
            gy[c * B + nn] = gyh[c * B + nn] * amp_mean / mabs[c] - dot * amp_mean / (mabs[c] * mabs[c]) * sg / (REAL)B;
        }
    }
    for (int k = 0; k < M; k++) {
        REAL g0 = 0, g1 = 0;
        for (int nn = 0; nn < B; nn++) {
            const int s = nn * sps + k - pad;
            if (s < 0 || s >= L) continue;
            g0 += gy[nn] * x[s] + gy[B + nn] * x[L + s];
            g1 += gy[nn] * x[L + s] - gy[B + nn] * x[s];
        }
        gW[k] = g0;
        gW[M + k] = g1;
    }
    free(buf);
    return loss;
}

/* R10: AWGN minibatch loop with Adam(amsgrad=True), AWGN:283,297-306 */
void FN(vaeq_oracle_awgn_train)(int n_steps, int B, int sps, int M, int n, int S, const REAL *rx, REAL *W, REAL *h,
                                REAL *mW, REAL *vW, REAL *vmaxW, REAL *mh_, REAL *vh, REAL *vmaxh, int *step,
                                const REAL *amp, const REAL *P, REAL amp_mean, REAL var, double lr, REAL *loss)
{
    const int L = B * sps;
    REAL *mb = (REAL *)malloc(sizeof(REAL) * (2 * L + 2 * n * B + 2 * B + 4 * M));
    REAL *q = mb + 2 * L, *out = q + 2 * n * B, *gW = out + 2 * B, *gh = gW + 2 * M;
    for (int s = 0; s < n_steps; s++) {
        for (int r = 0; r < 2; r++) memcpy(mb + r * L, rx + (size_t)r * S + s * L, sizeof(REAL) * L);   /* AWGN:299 */
        loss[s] = FN(vaeq_oracle_awgn_step_grads)(B, sps, M, n, mb, W, h, amp, P, amp_mean, var, q, out, gW, gh);
        *step += 1;
        FN(vaeq_oracle_adam)(2 * M, W, gW, mW, vW, vmaxW, *step, lr, 1);
        FN(vaeq_oracle_adam)(2 * M, h, gh, mh_, vh, vmaxh, *step, lr, 1);
    }
    free(mb);
}

/* ============================================================== row f3: AWGN VAE-NN (AWGN_channel/func_VAENN_MQAM.py)
 * Net (NN:170-188): fc1 = Conv1d(2, C, k1, pad k1/2), ELU, fc2 = Conv1d(C, C, k2, pad k2/2, stride sps), C = 2 n;
 * residual x_res = mean of the sps samples of a symbol added to every logit of its axis; per-axis softmax.
 * Parameters as one flat vector  theta = [w1[C][2][k1] | b1[C] | w2[C][C][k2] | b2[C] | h[2][M]]. */
#define NN_NP(C_, k1_, k2_, M_) ((C_) * 2 * (k1_) + (C_) + (C_) * (C_) * (k2_) + (C_) + 2 * (M_))

/* forward: x[2][L] -> z1[C][L] (ELU output), a2[C][B] (fc2 output, before the residual), q[C][B] */
void FN(vaeq_oracle_nn_forward)(int B, int sps, int n, int k1, int k2, const REAL *x, const REAL *theta, REAL *z1, REAL *a2, REAL *q)
{
    const int C = 2 * n, L = B * sps, p1 = k1 / 2, p2 = k2 / 2;
    const REAL *w1 = theta, *b1 = w1 + C * 2 * k1, *w2 = b1 + C, *b2 = w2 + C * C * k2;
    for (int c = 0; c < C; c++)
        for (int s = 0; s < L; s++) {
            REAL a = b1[c];
            for (int i = 0; i < 2; i++)
                for (int k = 0; k < k1; k++) {
                    const int sx = s + k - p1;
                    if (sx >= 0 && sx < L) a += w1[(c * 2 + i) * k1 + k] * x[i * L + sx];
                }
            z1[c * L + s] = a > 0 ? a : FN(r_exp)(a) - 1;        /* F.elu, alpha = 1 (NN:177) */
        }
    for (int c = 0; c < C; c++)
        for (int nn = 0; nn < B; nn++) {
            REAL a = b2[c];
            for (int cc = 0; cc < C; cc++)
                for (int k = 0; k < k2; k++) {
                    const int sx = nn * sps + k - p2;
                    if (sx >= 0 && sx < L) a += w2[(c * C + cc) * k2 + k] * z1[cc * L + sx];
                }
            a2[c * B + nn] = a;
        }
    for (int ax = 0; ax < 2; ax++)
        for (int nn = 0; nn < B; nn++) {
            REAL xres = 0;                                       /* NN:181-183 */
            for (int i = 0; i < sps; i++) xres += x[ax * L + nn * sps + i] / (REAL)sps;
            REAL lg[MAXLEV], mx = -1e30, sum = 0;
            for (int i = 0; i < n; i++) { lg[i] = a2[(ax * n + i) * B + nn] + xres; if (lg[i] > mx) mx = lg[i]; }
            for (int i = 0; i < n; i++) { lg[i] = FN(r_exp)(lg[i] - mx); sum += lg[i]; }
            for (int i = 0; i < n; i++) q[(ax * n + i) * B + nn] = lg[i] / sum;   /* nn.Softmax(dim=1), NN:184-186 */
        }
}

/* loss_function (NN:63-95): the AWGN VAE-LE ELBO without the prior (entropy instead of KL): awgn_loss_core with P = 1 */
REAL FN(vaeq_oracle_nn_loss)(int B, int sps, int M, int n, const REAL *q, const REAL *x, const REAL *h, const REAL *amp)
{
    REAL ones[MAXLEV];
    for (int i = 0; i < MAXLEV; i++) ones[i] = 1;
    return FN(vaeq_oracle_awgn_loss)(B, sps, M, n, q, x, h, amp, ones);
}

/* forward + loss + backward the way autograd walks it (NN:279-285): g = dL/dtheta (flat), q out */
REAL FN(vaeq_oracle_nn_step_grads)(int B, int sps, int M, int n, int k1, int k2, const REAL *x, const REAL *theta, const REAL *amp,
                                   REAL *q, REAL *g)
{
    const int C = 2 * n, L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh, p1 = k1 / 2, p2 = k2 / 2;
    const REAL *w1 = theta, *w2 = theta + C * 2 * k1 + C, *h = theta + NN_NP(C, k1, k2, M) - 2 * M;
    REAL *gw1 = g, *gb1 = gw1 + C * 2 * k1, *gw2 = gb1 + C, *gb2 = gw2 + C * C * k2, *gh = gb2 + C;
    (void)w1;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * ((size_t)2 * C * L + 2 * C * B + 8 * L + 2 * nm));
    REAL *z1 = buf, *gz = z1 + C * L, *a2 = gz + C * L, *ga2 = a2 + C * B;
    REAL *Eq = ga2 + C * B, *Eq2 = Eq + 2 * L, *gEq = Eq2 + 2 * L, *gEq2 = gEq + 2 * L, *Dre = gEq2 + 2 * L, *Dim = Dre + nm;
    REAL ones[MAXLEV], Cc;
    for (int i = 0; i < MAXLEV; i++) ones[i] = 1;
    FN(vaeq_oracle_nn_forward)(B, sps, n, k1, k2, x, theta, z1, a2, q);
    const REAL loss = FN(awgn_loss_core)(B, sps, M, n, q, x, h, amp, ones, Eq, Eq2, Dre, Dim, &Cc);
    const REAL gC = (REAL)nm / Cc;
    for (int i = 0; i < 2 * L; i++) { gEq[i] = 0; gEq2[i] = 0; }
    for (int j = 0; j <= Mh; j++) {                              /* same graph as the VAE-LE loss (NN:85-88) */
        const REAL hr = h[j], hi = h[M + j], hh = hr * hr + hi * hi;
        REAL ghr = 0, ghi = 0, vs = 0;
        for (int t = 0; t < nm; t++) {
            const int s = t + Mh - j;
            const REAL dDr = -2 * (x[mh + t] - Dre[t]) * gC, dDi = -2 * (x[L + mh + t] - Dim[t]) * gC;
            ghr += dDr * Eq[s] + dDi * Eq[L + s];
            ghi += -dDr * Eq[L + s] + dDi * Eq[s];
            gEq[s] += dDr * hr + dDi * hi;
            gEq[L + s] += -dDr * hi + dDi * hr;
            gEq2[s] += gC * hh;
            gEq2[L + s] += gC * hh;
            gEq[s] += gC * hh * (-2 * Eq[s]);
            gEq[L + s] += gC * hh * (-2 * Eq[L + s]);
            vs += (Eq2[s] - Eq[s] * Eq[s]) + (Eq2[L + s] - Eq[L + s] * Eq[L + s]);
        }
        gh[j] = ghr + gC * 2 * hr * vs;
        gh[M + j] = ghi + gC * 2 * hi * vs;
    }
    for (int ax = 0; ax < 2; ax++)                               /* dL/dq, softmax backward -> dL/da2 */
        for (int nn = 0; nn < B; nn++) {
            const REAL gmu = gEq[ax * L + nn * sps], grho = gEq2[ax * L + nn * sps];
            const int inr = (nn >= mh && nn < B - mh);
            REAL gq[MAXLEV], dot = 0;
            for (int i = 0; i < n; i++) {
                const REAL qq = q[(ax * n + i) * B + nn];
                gq[i] = amp[i] * gmu + (amp[i] * amp[i]) * grho;
                if (inr) gq[i] += FN(r_log)(qq + (REAL)1e-12) + qq / (qq + (REAL)1e-12);   /* d(q log(q+eps))/dq, NN:90 */
                dot += qq * gq[i];
            }
            for (int i = 0; i < n; i++) ga2[(ax * n + i) * B + nn] = q[(ax * n + i) * B + nn] * (gq[i] - dot);
        }
    for (int i = 0; i < C * L; i++) gz[i] = 0;
    for (int c = 0; c < C; c++) {                                /* fc2 backward */
        REAL sb = 0;
        for (int nn = 0; nn < B; nn++) sb += ga2[c * B + nn];
        gb2[c] = sb;
        for (int cc = 0; cc < C; cc++)
            for (int k = 0; k < k2; k++) {
                REAL sw = 0;
                const REAL w = w2[(c * C + cc) * k2 + k];
                for (int nn = 0; nn < B; nn++) {
                    const int sx = nn * sps + k - p2;
                    if (sx < 0 || sx >= L) continue;
                    sw += ga2[c * B + nn] * z1[cc * L + sx];
                    gz[cc * L + sx] += w * ga2[c * B + nn];
                }
                gw2[(c * C + cc) * k2 + k] = sw;
            }
    }
    for (int i = 0; i < C * L; i++) gz[i] *= z1[i] > 0 ? 1 : z1[i] + 1;   /* ELU': 1 or exp(a) = z1 + 1 */
    for (int c = 0; c < C; c++) {                                /* fc1 backward */
        REAL sb = 0;
        for (int s = 0; s < L; s++) sb += gz[c * L + s];
        gb1[c] = sb;
        for (int i = 0; i < 2; i++)
            for (int k = 0; k < k1; k++) {
                REAL sw = 0;
                for (int s = 0; s < L; s++) {
                    const int sx = s + k - p1;
                    if (sx >= 0 && sx < L) sw += gz[c * L + s] * x[i * L + sx];
                }
                gw1[(c * 2 + i) * k1 + k] = sw;
            }
    }
    free(buf);
    return loss;
}

/* minibatch loop with Adam(amsgrad=True) on every parameter (NN:248-253, 274-285) */
void FN(vaeq_oracle_nn_train)(int n_steps, int B, int sps, int M, int n, int k1, int k2, int S, const REAL *rx, REAL *theta, REAL *am,
                              REAL *av, REAL *avmax, int *step, const REAL *amp, double lr, REAL *loss)
{
    const int C = 2 * n, L = B * sps, np_ = NN_NP(C, k1, k2, M);
    REAL *mb = (REAL *)malloc(sizeof(REAL) * ((size_t)2 * L + C * B + np_));
    REAL *q = mb + 2 * L, *g = q + C * B;
    for (int s = 0; s < n_steps; s++) {
        for (int r = 0; r < 2; r++) memcpy(mb + r * L, rx + (size_t)r * S + (size_t)s * L, sizeof(REAL) * L);   /* NN:276 */
        loss[s] = FN(vaeq_oracle_nn_step_grads)(B, sps, M, n, k1, k2, mb, theta, amp, q, g);
        *step += 1;
        FN(vaeq_oracle_adam)(np_, theta, g, am, av, avmax, *step, lr, 1);
    }
    free(mb);
}

/* ---- Net_BN (NN:190-211): Net with BatchNorm1d(C) between the ELU and fc2, fc1 initialised with kaiming_uniform_.
 * theta = [w1 | b1 | w2 | b2 | gamma C | beta C | h 2M] (net.parameters() order, then h_est); bn = [running_mean C | running_var C].
 * training != 0: batch statistics over the L samples of the minibatch (biased variance, eps 1e-5), running stats updated with
 * momentum 0.1 and the unbiased variance; training == 0: running statistics (net.eval(), NN:287). */
#define NNBN_NP(C_, k1_, k2_, M_) (NN_NP(C_, k1_, k2_, M_) + 2 * (C_))
void FN(vaeq_oracle_nnbn_forward)(int B, int sps, int n, int k1, int k2, int training, const REAL *x, const REAL *theta, REAL *bn,
                                  REAL *zhat /*[C][L]*/, REAL *stat /*[2][C] mean, rstd*/, REAL *zb /*[C][L]*/, REAL *a2, REAL *q)
{
    const int C = 2 * n, L = B * sps, p1 = k1 / 2, p2 = k2 / 2;
    const REAL *w1 = theta, *b1 = w1 + C * 2 * k1, *w2 = b1 + C, *b2 = w2 + C * C * k2, *gam = b2 + C, *bet = gam + C;
    for (int c = 0; c < C; c++) {
        for (int s = 0; s < L; s++) {
            REAL a = b1[c];
            for (int i = 0; i < 2; i++)
                for (int k = 0; k < k1; k++) {
                    const int sx = s + k - p1;
                    if (sx >= 0 && sx < L) a += w1[(c * 2 + i) * k1 + k] * x[i * L + sx];
                }
            zhat[c * L + s] = a > 0 ? a : FN(r_exp)(a) - 1;      /* ELU output, normalised in place below */
        }
        REAL mean, rstd;
        if (training) {
            REAL sm = 0, sv = 0;
            for (int s = 0; s < L; s++) sm += zhat[c * L + s];
            mean = sm / (REAL)L;
            for (int s = 0; s < L; s++) sv += (zhat[c * L + s] - mean) * (zhat[c * L + s] - mean);
            const REAL var = sv / (REAL)L;
            rstd = 1 / FN(r_sqrt)(var + (REAL)1e-5);
            bn[c] = (REAL)0.9 * bn[c] + (REAL)0.1 * mean;
            bn[C + c] = (REAL)0.9 * bn[C + c] + (REAL)0.1 * (var * (REAL)L / (REAL)(L - 1));
        } else {
            mean = bn[c];
            rstd = 1 / FN(r_sqrt)(bn[C + c] + (REAL)1e-5);
        }
        stat[c] = mean; stat[C + c] = rstd;
        for (int s = 0; s < L; s++) {
            zhat[c * L + s] = (zhat[c * L + s] - mean) * rstd;
            zb[c * L + s] = gam[c] * zhat[c * L + s] + bet[c];
        }
    }
    for (int c = 0; c < C; c++)
        for (int nn = 0; nn < B; nn++) {
            REAL a = b2[c];
            for (int cc = 0; cc < C; cc++)
                for (int k = 0; k < k2; k++) {
                    const int sx = nn * sps + k - p2;
                    if (sx >= 0 && sx < L) a += w2[(c * C + cc) * k2 + k] * zb[cc * L + sx];
                }
            a2[c * B + nn] = a;
        }
    for (int ax = 0; ax < 2; ax++)
        for (int nn = 0; nn < B; nn++) {
            REAL xres = 0;
            for (int i = 0; i < sps; i++) xres += x[ax * L + nn * sps + i] / (REAL)sps;
            REAL lg[MAXLEV], mx = -1e30, sum = 0;
            for (int i = 0; i < n; i++) { lg[i] = a2[(ax * n + i) * B + nn] + xres; if (lg[i] > mx) mx = lg[i]; }
            for (int i = 0; i < n; i++) { lg[i] = FN(r_exp)(lg[i] - mx); sum += lg[i]; }
            for (int i = 0; i < n; i++) q[(ax * n + i) * B + nn] = lg[i] / sum;
        }
}

REAL FN(vaeq_oracle_nnbn_step_grads)(int B, int sps, int M, int n, int k1, int k2, const REAL *x, const REAL *theta, REAL *bn,
                                     const REAL *amp, REAL *q, REAL *g)
{
    const int C = 2 * n, L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh, p1 = k1 / 2, p2 = k2 / 2;
    const REAL *w2 = theta + C * 2 * k1 + C, *gam = w2 + C * C * k2 + C, *h = theta + NNBN_NP(C, k1, k2, M) - 2 * M;
    REAL *gw1 = g, *gb1 = gw1 + C * 2 * k1, *gw2 = gb1 + C, *gb2 = gw2 + C * C * k2, *gga = gb2 + C, *gbe = gga + C, *gh = gbe + C;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * ((size_t)3 * C * L + 2 * C + 2 * C * B + 8 * L + 2 * nm));
    REAL *zhat = buf, *zb = zhat + C * L, *gz = zb + C * L, *stat = gz + C * L, *a2 = stat + 2 * C, *ga2 = a2 + C * B;
    REAL *Eq = ga2 + C * B, *Eq2 = Eq + 2 * L, *gEq = Eq2 + 2 * L, *gEq2 = gEq + 2 * L, *Dre = gEq2 + 2 * L, *Dim = Dre + nm;
    REAL ones[MAXLEV], Cc;
    for (int i = 0; i < MAXLEV; i++) ones[i] = 1;
    FN(vaeq_oracle_nnbn_forward)(B, sps, n, k1, k2, 1, x, theta, bn, zhat, stat, zb, a2, q);
    const REAL loss = FN(awgn_loss_core)(B, sps, M, n, q, x, h, amp, ones, Eq, Eq2, Dre, Dim, &Cc);
    const REAL gC = (REAL)nm / Cc;
    for (int i = 0; i < 2 * L; i++) { gEq[i] = 0; gEq2[i] = 0; }
    for (int j = 0; j <= Mh; j++) {
        const REAL hr = h[j], hi = h[M + j], hh = hr * hr + hi * hi;
        REAL ghr = 0, ghi = 0, vs = 0;
        for (int t = 0; t < nm; t++) {
            const int s = t + Mh - j;
            const REAL dDr = -2 * (x[mh + t] - Dre[t]) * gC, dDi = -2 * (x[L + mh + t] - Dim[t]) * gC;
            ghr += dDr * Eq[s] + dDi * Eq[L + s];
            ghi += -dDr * Eq[L + s] + dDi * Eq[s];
            gEq[s] += dDr * hr + dDi * hi;
            gEq[L + s] += -dDr * hi + dDi * hr;
            gEq2[s] += gC * hh;
            gEq2[L + s] += gC * hh;
            gEq[s] += gC * hh * (-2 * Eq[s]);
            gEq[L + s] += gC * hh * (-2 * Eq[L + s]);
            vs += (Eq2[s] - Eq[s] * Eq[s]) + (Eq2[L + s] - Eq[L + s] * Eq[L + s]);
        }
        gh[j] = ghr + gC * 2 * hr * vs;
        gh[M + j] = ghi + gC * 2 * hi * vs;
    }
    for (int ax = 0; ax < 2; ax++)
        for (int nn = 0; nn < B; nn++) {
            const REAL gmu = gEq[ax * L + nn * sps], grho = gEq2[ax * L + nn * sps];
            const int inr = (nn >= mh && nn < B - mh);
            REAL gq[MAXLEV], dot = 0;
            for (int i = 0; i < n; i++) {
                const REAL qq = q[(ax * n + i) * B + nn];
                gq[i] = amp[i] * gmu + (amp[i] * amp[i]) * grho;
                if (inr) gq[i] += FN(r_log)(qq + (REAL)1e-12) + qq / (qq + (REAL)1e-12);
                dot += qq * gq[i];
            }
            for (int i = 0; i < n; i++) ga2[(ax * n + i) * B + nn] = q[(ax * n + i) * B + nn] * (gq[i] - dot);
        }
    for (int i = 0; i < C * L; i++) gz[i] = 0;
    for (int c = 0; c < C; c++) {                                /* fc2 backward (input = BN output zb) */
        REAL sb = 0;
        for (int nn = 0; nn < B; nn++) sb += ga2[c * B + nn];
        gb2[c] = sb;
        for (int cc = 0; cc < C; cc++)
            for (int k = 0; k < k2; k++) {
                REAL sw = 0;
                const REAL w = w2[(c * C + cc) * k2 + k];
                for (int nn = 0; nn < B; nn++) {
                    const int sx = nn * sps + k - p2;
                    if (sx < 0 || sx >= L) continue;
                    sw += ga2[c * B + nn] * zb[cc * L + sx];
                    gz[cc * L + sx] += w * ga2[c * B + nn];
                }
                gw2[(c * C + cc) * k2 + k] = sw;
            }
    }
    for (int c = 0; c < C; c++) {                                /* BatchNorm backward (batch statistics), then ELU' */
        REAL s1 = 0, s2 = 0;
        for (int s = 0; s < L; s++) { s1 += gz[c * L + s]; s2 += gz[c * L + s] * zhat[c * L + s]; }
        gbe[c] = s1;
        gga[c] = s2;
        const REAL mean = stat[c], rstd = stat[C + c];
        for (int s = 0; s < L; s++) {
            const REAL zh = zhat[c * L + s];
            const REAL gzz = gam[c] * rstd * (gz[c * L + s] - s1 / (REAL)L - zh * s2 / (REAL)L);
            const REAL z = zh / rstd + mean;                     /* ELU output before the normalisation */
            gz[c * L + s] = gzz * (z > 0 ? 1 : z + 1);
        }
    }
    for (int c = 0; c < C; c++) {                                /* fc1 backward */
        REAL sb = 0;
        for (int s = 0; s < L; s++) sb += gz[c * L + s];
        gb1[c] = sb;
        for (int i = 0; i < 2; i++)
            for (int k = 0; k < k1; k++) {
                REAL sw = 0;
                for (int s = 0; s < L; s++) {
                    const int sx = s + k - p1;
                    if (sx >= 0 && sx < L) sw += gz[c * L + s] * x[i * L + sx];
                }
                gw1[(c * 2 + i) * k1 + k] = sw;
            }
    }
    free(buf);
    return loss;
}

void FN(vaeq_oracle_nnbn_train)(int n_steps, int B, int sps, int M, int n, int k1, int k2, int S, const REAL *rx, REAL *theta, REAL *bn,
                                REAL *am, REAL *av, REAL *avmax, int *step, const REAL *amp, double lr, REAL *loss)
{
    const int C = 2 * n, L = B * sps, np_ = NNBN_NP(C, k1, k2, M);
    REAL *mb = (REAL *)malloc(sizeof(REAL) * ((size_t)2 * L + C * B + np_));
    REAL *q = mb + 2 * L, *g = q + C * B;
    for (int s = 0; s < n_steps; s++) {
        for (int r = 0; r < 2; r++) memcpy(mb + r * L, rx + (size_t)r * S + (size_t)s * L, sizeof(REAL) * L);
        loss[s] = FN(vaeq_oracle_nnbn_step_grads)(B, sps, M, n, k1, k2, mb, theta, bn, amp, q, g);
        *step += 1;
        FN(vaeq_oracle_adam)(np_, theta, g, am, av, avmax, *step, lr, 1);
    }
    free(mb);
}

/* eval-mode forward (net.eval(): running statistics) */
void FN(vaeq_oracle_nnbn_forward_eval)(int B, int sps, int n, int k1, int k2, const REAL *x, const REAL *theta, const REAL *bn, REAL *q)
{
    const int C = 2 * n, L = B * sps;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * ((size_t)2 * C * L + 2 * C + C * B));
    REAL bnc[2 * 2 * MAXLEV];
    for (int i = 0; i < 2 * C; i++) bnc[i] = bn[i];
    FN(vaeq_oracle_nnbn_forward)(B, sps, n, k1, k2, 0, x, theta, bnc, buf, buf + 2 * C * L, buf + C * L, buf + 2 * C * L + 2 * C, q);
    free(buf);
}
#undef NNBN_NP
#undef NN_NP

/* ============================================================== row f4: constant-modulus baselines (shared_funcs.py:341-488)
 * CMA (:341-383), CMAbatch (:385-433), CMAflex (:435-488) on one frame Rx[2][2][N]:
 *   y = zero-padded Rx divided by the mean of |y_pol|^2 over the PADDED length (:350-351, a power -- not an amplitude -- scaling);
 *   symbol j (sample i = mh + sps j) is written to index k = i / sps - mh, which is NEGATIVE for the first symbols and then wraps to
 *   the end of the arrays exactly like the reference's tensor indexing does (:357);
 *   mode 0: tap update after every symbol (:371-381); mode 1: the increments of the last `batchlen` symbols are applied when
 *   k % symb_step == 0 and k >= batchlen (CMAflex :475; CMAbatch = the same with symb_step = batchlen, :421).
 * h[2][2][2][M] = [out pol][in pol][re/im][tap], updated in place; out[2][2][K], e[K][2], K = N / sps. */
void FN(vaeq_oracle_cma)(int N, int sps, int M, int mode, int batchlen, int symb_step, const REAL *rx, REAL R, REAL *h, double lr,
                         REAL *out, REAL *e)
{
    const int mh = M / 2, Lp = N + 2 * mh, K = N / sps;
    REAL *y = (REAL *)calloc((size_t)4 * Lp, sizeof(REAL));
    REAL *gbuf = (REAL *)calloc((size_t)K * 4 + 1, sizeof(REAL));      /* per symbol: out[0][0], out[0][1], out[1][0], out[1][1] at update time */
    int *jof = (int *)malloc(sizeof(int) * (K + 1));                   /* symbol number j stored at index k (to find its window again) */
    REAL pw = 0;
    for (int p = 0; p < 2; p++)
        for (int c = 0; c < 2; c++)
            for (int i = 0; i < N; i++) {
                const REAL v = rx[(p * 2 + c) * N + i];
                y[(p * 2 + c) * Lp + mh + i] = v;
                pw += v * v;
            }
    pw /= (REAL)(2 * Lp);                                              /* torch.mean over [2, Lp] */
    for (int i = 0; i < 4 * Lp; i++) y[i] /= pw;
    for (int i = 0; i < 4 * K; i++) out[i] = 0;
    const REAL two_lr = (REAL)(2.0 * lr);
    for (int j = 0; mh + sps * j < N + mh; j++) {
        const int i0 = sps * j;                                        /* first padded sample of the window: i - mh */
        int k = (mh + sps * j) / sps - mh;
        const int kraw = k;
        if (k < 0) k += K;
        REAL o[2][2];
        for (int op = 0; op < 2; op++) {
            REAL re = 0, im = 0;
            for (int p = 0; p < 2; p++) {
                REAL s0 = 0, s1 = 0, s2 = 0, s3 = 0;                   /* torch.matmul dot products, accumulated per filter row */
                for (int t = 0; t < M; t++) {
                    const REAL yr = y[(p * 2 + 0) * Lp + i0 + t], yi = y[(p * 2 + 1) * Lp + i0 + t];
                    const REAL hr = h[((op * 2 + p) * 2 + 0) * M + t], hi = h[((op * 2 + p) * 2 + 1) * M + t];
                    s0 += yr * hr; s1 += yi * hi; s2 += yr * hi; s3 += yi * hr;
                }
                re += s0 - s1;
                im += s2 + s3;
            }
            o[op][0] = re; o[op][1] = im;
            out[(op * 2 + 0) * K + k] = re;
            out[(op * 2 + 1) * K + k] = im;
            e[k * 2 + op] = R - re * re - im * im;
        }
        if (mode == 0) {
            for (int op = 0; op < 2; op++)
                for (int p = 0; p < 2; p++)
                    for (int t = 0; t < M; t++) {
                        const REAL yr = y[(p * 2 + 0) * Lp + i0 + t], yi = y[(p * 2 + 1) * Lp + i0 + t], ee = e[k * 2 + op];
                        h[((op * 2 + p) * 2 + 0) * M + t] += two_lr * ee * (o[op][0] * yr + o[op][1] * yi);
                        h[((op * 2 + p) * 2 + 1) * M + t] += two_lr * ee * (o[op][1] * yr - o[op][0] * yi);
                    }
        } else {
            gbuf[k * 4 + 0] = o[0][0]; gbuf[k * 4 + 1] = o[0][1]; gbuf[k * 4 + 2] = o[1][0]; gbuf[k * 4 + 3] = o[1][1];
            jof[k] = j;
            if (kraw >= batchlen && kraw % symb_step == 0) {
                for (int op = 0; op < 2; op++)
                    for (int p = 0; p < 2; p++)
                        for (int t = 0; t < M; t++) {
                            REAL a0 = 0, a1 = 0;
                            for (int kk = kraw - batchlen; kk < kraw; kk++) {
                                const int w0 = sps * jof[kk];
                                const REAL yr = y[(p * 2 + 0) * Lp + w0 + t], yi = y[(p * 2 + 1) * Lp + w0 + t], ee = e[kk * 2 + op];
                                a0 += (gbuf[kk * 4 + op * 2] * yr + gbuf[kk * 4 + op * 2 + 1] * yi) * ee;
                                a1 += (gbuf[kk * 4 + op * 2 + 1] * yr - gbuf[kk * 4 + op * 2] * yi) * ee;
                            }
                            h[((op * 2 + p) * 2 + 0) * M + t] += two_lr * a0;
                            h[((op * 2 + p) * 2 + 1) * M + t] += two_lr * a1;
                        }
            }
        }
    }
    free(y); free(gbuf); free(jof);
}

#undef FN
#undef CAT
#undef CAT_
#undef MAXLEV
