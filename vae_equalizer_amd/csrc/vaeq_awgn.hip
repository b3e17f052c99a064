// vaeq_awgn.hip -- single-polarisation (AWGN / ISI channel) VAE-LE training loop for gfx950.
//
// Same structure as vaeq_dp.hip (one workgroup = one run, everything in LDS) for the 1x1 variant of
// AWGN_channel/func_VAELE_MQAM_shaping.py:
//   twoFIR.forward (:214-231)   y = (W0 - j W1) * x, pad (M-1)/2, stride sps;
//                               yhat_c = y_c / mean_n|y_c| * amp_mean  (:228, differentiable);
//                               q_i = softmax_i( -(yhat_c - a_i)^2 / var )  -- no 1/2, no PCS term (:229)
//   loss_function (:63-95)      C = sum|x - D|^2 + sum_j |h_j|^2 VS[j];  loss = nm log C + sum q log(q/P + 1e-12)
//   Adam(amsgrad=True) (:283)   one learning rate for both groups, no schedule
// Backward = vaeq_dp.hip's closed form specialised to one polarisation, with dz_i/dyhat = -2 (yhat - a_i) / var and
// the normalisation's Jacobian:
//   dL/dy_c[n] = g_c[n] A/m_c - (sum_n' g_c[n'] y_c[n']) A / m_c^2 * sign(y_c[n]) / B,   g = dL/dyhat, m_c = mean|y_c|
//   dL/dW0[k] = sum_n gI x0 + gQ x1,   dL/dW1[k] = sum_n gI x1 - gQ x0        (w = W0 - j W1)
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vaeq.h"
#include "vaeq_common.h"
#include "vaeq_noise.h"
#include "vaeq_wave.h"
#include "vaeq_validate.h"

namespace vaeq {

// wave-per-run fast path (vaeq_awgn_wave.hip)
bool awgn_wave_supported(const vaeq_awgn_args &a);
int launch_awgn_wave(const vaeq_awgn_args &a, hipStream_t st);

struct AWGNLayout {
    int L, mh, Mh, nm, Lp;
    int xs, Ws, hs, mW, vW, xW, mH, vH, xH, gW, gH, ys, mu, vr, t3, kc, gy, es, VS, red, total;
};

__host__ __device__ inline int apad4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline AWGNLayout awgn_layout(int B, int sps, int M)
{
    AWGNLayout l;
    l.L = B * sps;
    l.mh = M / 2;
    l.Mh = 2 * l.mh;
    l.nm = l.L - l.Mh;
    l.Lp = apad4(l.L + 2 * l.mh);
    int o = 0;
    auto take = [&](int n) { int r = o; o += apad4(n); return r; };
    l.xs = take(2 * l.Lp);
    l.Ws = take(2 * M); l.hs = take(2 * M);
    l.mW = take(2 * M); l.vW = take(2 * M); l.xW = take(2 * M);
    l.mH = take(2 * M); l.vH = take(2 * M); l.xH = take(2 * M);
    l.gW = take(2 * M); l.gH = take(2 * M);
    l.ys = take(2 * B);
    l.mu = take(2 * B); l.vr = take(2 * B); l.t3 = take(2 * B); l.kc = take(2 * B); l.gy = take(2 * B);
    l.es = take(2 * l.nm);
    l.VS = take(M);
    l.red = take(64);
    l.total = o;
    return l;
}

template <int NT, int NLEV>
__global__ __launch_bounds__(NT) void awgn_train_kernel(const vaeq_awgn_args a)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    const int tid = threadIdx.x, run = blockIdx.x;
    const int B = a.B, sps = a.sps, M = a.M;
    const AWGNLayout l = awgn_layout(B, sps, M);
    const int L = l.L, mh = l.mh, Mh = l.Mh, nm = l.nm, Lp = l.Lp;
    float *xs = sm + l.xs, *Ws = sm + l.Ws, *hs = sm + l.hs;
    float *mWs = sm + l.mW, *vWs = sm + l.vW, *xWs = sm + l.xW, *mHs = sm + l.mH, *vHs = sm + l.vH, *xHs = sm + l.xH;
    float *gWs = sm + l.gW, *gHs = sm + l.gH, *ys = sm + l.ys;
    float *mu = sm + l.mu, *vr = sm + l.vr, *t3 = sm + l.t3, *kc = sm + l.kc, *gy = sm + l.gy, *es = sm + l.es;
    float *VS = sm + l.VS, *red = sm + l.red;
    const int NP = 2 * M;

    float amp[NLEV], invP[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) {
        amp[i] = a.amp[i];
        invP[i] = 1.0f / a.P[(size_t)run * NLEV + i];
    }
    const float A = a.amp_mean[run], var = a.var[run], ivar = 1.0f / var;
    const double lr = (double)a.lr[run];

    for (int i = tid; i < NP; i += NT) {
        const size_t g = (size_t)run * NP + i;
        Ws[i] = a.W[g]; hs[i] = a.h[g];
        mWs[i] = a.adam_mW[g]; vWs[i] = a.adam_vW[g]; xWs[i] = a.adam_xW[g];
        mHs[i] = a.adam_mh[g]; vHs[i] = a.adam_vh[g]; xHs[i] = a.adam_xh[g];
    }
    int step = a.step[run];
    double b1t = pow(0.9, (double)step), b2t = pow(0.999, (double)step);
    __syncthreads();

    const size_t No = (size_t)a.steps * B;
    const float *rxr = a.rx + (size_t)run * 2 * (size_t)a.S;
    float *qf = a.q_out ? a.q_out + (size_t)run * 2 * NLEV * No : nullptr;
    float *yf = a.y_out ? a.y_out + (size_t)run * 2 * No : nullptr;

    for (int s = 0; s < a.steps; s++) {
        // ---- P0: minibatch -> LDS with zero halo (:299, Conv1d padding :209)
        const size_t s0 = (size_t)s * L;
        for (int i = tid; i < 2 * Lp; i += NT) {
            const int row = i / Lp, c = i - row * Lp, sx = c - mh;
            xs[i] = (sx >= 0 && sx < L) ? rxr[(size_t)row * a.S + s0 + sx] : 0.0f;
        }
        __syncthreads();

        // ---- P1a: FIR, |y| sums
        float sa0 = 0.f, sa1 = 0.f;
        for (int n = tid; n < B; n += NT) {
            const float *x0 = xs + n * sps, *x1 = xs + Lp + n * sps;
            float yI = 0.f, yQ = 0.f;
            for (int k = 0; k < M; k++) {
                const float a_ = x0[k], b_ = x1[k], c_ = Ws[k], d_ = Ws[M + k];
                yI = fmaf(c_, a_, yI); yI = fmaf(d_, b_, yI);
                yQ = fmaf(c_, b_, yQ); yQ = fmaf(-d_, a_, yQ);
            }
            ys[n] = yI; ys[B + n] = yQ;
            if (yf) { yf[s * (size_t)B + n] = yI; yf[No + s * (size_t)B + n] = yQ; }   // un-normalised out (:227,231)
            sa0 += fabsf(yI); sa1 += fabsf(yQ);
        }
        block_reduce3<NT>(sa0, sa1, 0.f, red);
        const float m0 = red[0] / (float)B, m1 = red[1] / (float)B;   // mean|y_c| (:228)
        __syncthreads();                                               // red is reused below

        // ---- P1b: normalise, demap, moments; item = (c, n)
        float klsum = 0.f;
        for (int it = tid; it < 2 * B; it += NT) {
            const int c = it / B, n = it - c * B;
            const float y = ys[it] / (c ? m1 : m0) * A;
            const bool inr = (n >= mh) && (n < B - mh);
            float z[NLEV], zmax = -3.0e38f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) {
                const float d = y - amp[i];
                z[i] = -(d * d * ivar);
                zmax = fmaxf(zmax, z[i]);
            }
            float ssum = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] = __expf(z[i] - zmax); ssum += z[i]; }
            const float rs = 1.0f / ssum;
            float e1 = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] *= rs; e1 = fmaf(amp[i], z[i], e1); }
            if (qf) {
#pragma unroll
                for (int i = 0; i < NLEV; i++) qf[(size_t)(c * NLEV + i) * No + s * (size_t)B + n] = z[i];
            }
            float e2 = 0.f, e3 = 0.f, kk = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) {
                const float d = amp[i] - e1, qd = z[i] * d;
                e2 = fmaf(qd, d, e2);
                e3 = fmaf(qd * d, d, e3);
                if (inr) {
                    const float r = z[i] * invP[i], re = r + 1e-12f;
                    const float lg = __logf(re);
                    klsum = fmaf(z[i], lg, klsum);
                    kk = fmaf(qd, lg + r / re, kk);
                }
            }
            mu[it] = e1; vr[it] = e2; t3[it] = e3; kc[it] = kk;
        }
        __syncthreads();

        // ---- P2: e = x - D (item t), VS (item j)
        float se = 0.f;
        for (int t = tid; t < nm; t += NT) {
            float dr = 0.f, di = 0.f;
            for (int j = (t + Mh) % sps; j <= Mh; j += sps) {
                const int np = (t + Mh - j) / sps;
                const float a_ = mu[np], b_ = mu[B + np], c_ = hs[j], d_ = hs[M + j];
                dr = fmaf(c_, a_, dr); dr = fmaf(-d_, b_, dr);
                di = fmaf(c_, b_, di); di = fmaf(d_, a_, di);
            }
            const float er = xs[Mh + t] - dr, ei = xs[Lp + Mh + t] - di;
            es[t] = er; es[nm + t] = ei;
            se += er * er + ei * ei;
        }
        for (int j = tid; j < M; j += NT) {
            const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
            float acc = 0.f;
            for (int np = lo; np <= hi_; np++) acc += vr[np] + vr[B + np];
            VS[j] = acc;
        }
        block_reduce3<NT>(se, klsum, 0.f, red);
        float C = red[0];
        for (int j = 0; j < M; j++) C = fmaf(hs[j] * hs[j] + hs[M + j] * hs[M + j], VS[j], C);
        const float gC = (float)nm / C;
        if (tid == 0 && a.loss) a.loss[(size_t)run * a.steps + s] = (float)nm * logf(C) + red[1];

        // ---- P4a: dL/dh (item j)
        for (int j = tid; j < M; j += NT) {
            const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
            float ar = 0.f, ai = 0.f;
            for (int np = lo; np <= hi_; np++) {
                const int t = np * sps - Mh + j;
                const float a_ = es[t], b_ = es[nm + t], c_ = mu[np], d_ = mu[B + np];
                ar = fmaf(a_, c_, ar); ar = fmaf(b_, d_, ar);
                ai = fmaf(b_, c_, ai); ai = fmaf(-a_, d_, ai);
            }
            gHs[j] = gC * (-2.0f * ar + 2.0f * hs[j] * VS[j]);
            gHs[M + j] = gC * (-2.0f * ai + 2.0f * hs[M + j] * VS[j]);
        }
        // ---- P4b: dL/dU, G_V -> dL/dyhat (item n); dot_c = sum_n g_c y_c
        float dt0 = 0.f, dt1 = 0.f;
        for (int n = tid; n < B; n += NT) {
            const int sx = n * sps;
            const int jlo = max(0, Mh - sx), jhi = min(Mh, nm - 1 + Mh - sx);
            const float *er = es + (sx - Mh), *ei = er + nm;
            float pr = 0.f, pi = 0.f, ph = 0.f;
            for (int j = jlo; j <= jhi; j++) {
                const float a_ = er[j], b_ = ei[j], c_ = hs[j], d_ = hs[M + j];
                pr = fmaf(a_, c_, pr); pr = fmaf(b_, d_, pr);
                pi = fmaf(b_, c_, pi); pi = fmaf(-a_, d_, pi);
                ph = fmaf(c_, c_, ph); ph = fmaf(d_, d_, ph);
            }
            const float ur = -2.0f * gC * pr, ui = -2.0f * gC * pi, gv = gC * ph;
            const float gI = 2.0f * ivar * (ur * vr[n] + gv * t3[n] + kc[n]);
            const float gQ = 2.0f * ivar * (ui * vr[B + n] + gv * t3[B + n] + kc[B + n]);
            gy[n] = gI; gy[B + n] = gQ;
            dt0 = fmaf(gI, ys[n], dt0);
            dt1 = fmaf(gQ, ys[B + n], dt1);
        }
        __syncthreads();   // every thread has read red[0..1] of the previous reduction
        block_reduce3<NT>(dt0, dt1, 0.f, red);
        {
            const float s0_ = A / m0, s1_ = A / m1;
            const float k0_ = red[0] * A / (m0 * m0) / (float)B, k1_ = red[1] * A / (m1 * m1) / (float)B;
            __syncthreads();
            for (int n = tid; n < B; n += NT) {   // normalisation backward (:228)
                const float yI = ys[n], yQ = ys[B + n];
                const float sgI = (yI > 0.f) - (yI < 0.f), sgQ = (yQ > 0.f) - (yQ < 0.f);
                gy[n] = gy[n] * s0_ - k0_ * sgI;
                gy[B + n] = gy[B + n] * s1_ - k1_ * sgQ;
            }
        }
        __syncthreads();

        // ---- P5: dL/dW (item k)
        for (int k = tid; k < M; k += NT) {
            const float *x0 = xs + k, *x1 = xs + Lp + k;
            float g0 = 0.f, g1 = 0.f;
            for (int n = 0; n < B; n++) {
                const float a_ = gy[n], b_ = gy[B + n], c_ = x0[n * sps], d_ = x1[n * sps];
                g0 = fmaf(a_, c_, g0); g0 = fmaf(b_, d_, g0);
                g1 = fmaf(a_, d_, g1); g1 = fmaf(-b_, c_, g1);
            }
            gWs[k] = g0; gWs[M + k] = g1;
        }
        __syncthreads();

        // ---- P6: Adam(amsgrad)
        step += 1;
        b1t *= 0.9;
        b2t *= 0.999;
        if (!a.no_update) {
            const float bc2s = (float)sqrt(1.0 - b2t), ss = (float)(lr / (1.0 - b1t));
            for (int i = tid; i < 2 * NP; i += NT) {
                if (i < NP) adam_update_amsgrad(Ws[i], mWs[i], vWs[i], xWs[i], gWs[i], ss, bc2s);
                else adam_update_amsgrad(hs[i - NP], mHs[i - NP], vHs[i - NP], xHs[i - NP], gHs[i - NP], ss, bc2s);
            }
        }
        __syncthreads();
    }

    for (int i = tid; i < NP; i += NT) {
        const size_t g = (size_t)run * NP + i;
        if (!a.no_update) {
            a.W[g] = Ws[i]; a.h[g] = hs[i];
            a.adam_mW[g] = mWs[i]; a.adam_vW[g] = vWs[i]; a.adam_xW[g] = xWs[i];
            a.adam_mh[g] = mHs[i]; a.adam_vh[g] = vHs[i]; a.adam_xh[g] = xHs[i];
        }
        if (a.dbg_gW) a.dbg_gW[g] = gWs[i];
        if (a.dbg_gh) a.dbg_gh[g] = gHs[i];
    }
    if (tid == 0 && !a.no_update) a.step[run] = step;
}

// twoFIR.forward in eval mode on N symbols (validation, :311-313): one workgroup per run, two passes over y.
template <int NLEV>
__global__ __launch_bounds__(256) void awgn_forward_kernel(int64_t N, int sps, int M, const float *__restrict__ x, const float *__restrict__ W,
                                                           const float *__restrict__ amp_g, const float *__restrict__ amp_mean,
                                                           const float *__restrict__ var, float *__restrict__ q, float *__restrict__ yout)
{
    __shared__ float Ws[2 * 64];
    __shared__ float red[64];
    const int run = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2 * M; i += 256) Ws[i] = W[(size_t)run * 2 * M + i];
    __syncthreads();
    const int64_t L = N * sps;
    const int pad = (M - 1) / 2;
    const float *x0 = x + (size_t)run * 2 * L, *x1 = x0 + L;
    float *y0 = yout + (size_t)run * 2 * N, *y1 = y0 + N;
    float sa0 = 0.f, sa1 = 0.f;
    for (int64_t n = tid; n < N; n += 256) {
        float yI = 0.f, yQ = 0.f;
        for (int k = 0; k < M; k++) {
            const int64_t s = n * sps + k - pad;
            if (s < 0 || s >= L) continue;
            const float a_ = x0[s], b_ = x1[s];
            yI = fmaf(Ws[k], a_, yI); yI = fmaf(Ws[M + k], b_, yI);
            yQ = fmaf(Ws[k], b_, yQ); yQ = fmaf(-Ws[M + k], a_, yQ);
        }
        y0[n] = yI; y1[n] = yQ;
        sa0 += fabsf(yI); sa1 += fabsf(yQ);
    }
    block_reduce3<256>(sa0, sa1, 0.f, red);
    if (!q) return;
    float amp[NLEV], amp2[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; amp2[i] = 0.f; }
    const float A = amp_mean[run], ivar = 1.0f / var[run];
    const float s0 = A / (red[0] / (float)N), s1 = A / (red[1] / (float)N);
    for (int64_t it = tid; it < 2 * N; it += 256) {      // each thread re-reads only the y it wrote? no: any -> fence below
        const int c = it >= N;
        const int64_t n = it - (c ? N : 0);
        float qq[NLEV];
        // soft_demap computes -(d^2 * i2v + nusc*a^2): i2v = 1/var, nusc = 0 gives (yhat-a)^2/var (:229)
        soft_demap<NLEV>((c ? y1[n] : y0[n]) * (c ? s1 : s0), amp, amp2, ivar, 0.f, qq);
#pragma unroll
        for (int i = 0; i < NLEV; i++) q[((size_t)run * 2 * NLEV + c * NLEV + i) * N + n] = qq[i];
    }
}


// Fused validation pass of one epoch (func_VAELE_MQAM_shaping.py:308-318) for all runs, one workgroup per run:
//   twoFIR.forward in eval mode on N symbols (:311-313)  -> y (workspace; also the un-normalised output), mean |y| per axis
//   hard decisions = argmax_i q_i = nearest level of yhat (q itself is never materialised: 16 floats per symbol saved)
//   find_shift (:188-204): E_q[x_I] of the first 1000 symbols against the TX I (else Q) row over n_shift circular lags
//   SER_q (:97-123) on q[:, 11+sh : -11] vs data[:, 11 : -11-sh], minimum over the four quadrant rotations
// MT > 0: tap count baked in, four symbols per thread from one register window (sps == 2).
// GEN (MT > 0 only): x is not read but made -- the clean frame of vaeq_gen_awgn_clean plus the noise vaeq_gen_awgn would have added (vaeq_noise.h:
// same Philox words, same sigma_n, same fused multiply-adds), four samples = two noise words per staged chunk.  The 15 000-symbol validation frame
// of an epoch is then written once (clean) and read once instead of computed twice, written noisy and read (vaeq_awgn_validate_gen).
struct ValGen {
    const float2 *sig;           // [R][Ls] clean samples
    const float *part;           // [R][n_parts] tile power sums
    const float *snr_db;         // [R]
    const float *sigma_fixed;    // nullable [R]
    float *sigma_out;            // nullable [R]
    uint64_t seed;
    uint32_t frame;
    int Ls, n_parts;
};

template <int NLEV, int MT, bool GEN = false>
__global__ __launch_bounds__(256, 4) void awgn_validate_kernel(int N, int sps, int Mrt, int n_shift, const float *__restrict__ x,
                                                            const float *__restrict__ W, const float *__restrict__ amp_g,
                                                            const float *__restrict__ amp_mean, const float *__restrict__ var,
                                                            const __half *__restrict__ data, float *__restrict__ yws, float *__restrict__ ser_out,
                                                            int *__restrict__ shift_out, ValGen vg)
{
    extern __shared__ unsigned char decs[];            // [N] level decisions, I in the low and Q in the high nibble
    __shared__ float Ws[2 * 64];
    __shared__ float red[64];
    __shared__ float E[VAL_NE];
    __shared__ float corr[2][VAL_MAXSHIFT];
    __shared__ int sh_s;
    const int run = blockIdx.x, tid = threadIdx.x;
    const int M = MT ? MT : Mrt;
    for (int i = tid; i < 2 * M; i += 256) Ws[i] = W[(size_t)run * 2 * M + i];
    __syncthreads();
    const int64_t L = (int64_t)N * sps;
    const int pad = (M - 1) / 2;
    const float *x0 = GEN ? nullptr : x + (size_t)run * 2 * L, *x1 = GEN ? nullptr : x0 + L;
    float *y0 = yws + (size_t)run * 2 * N, *y1 = y0 + N;
    float sa0 = 0.f, sa1 = 0.f;
    float sigma = 0.f;
    const float2 *cs = nullptr;
    if (GEN) {
        sigma = vg.sigma_fixed ? vg.sigma_fixed[run] : awgn_sigma_from_parts(vg.part + (size_t)run * vg.n_parts, vg.n_parts, vg.Ls, sps, vg.snr_db[run]);
        if (vg.sigma_out && tid == 0) vg.sigma_out[run] = sigma;
        cs = vg.sig + (size_t)run * vg.Ls;
    }
    if (MT) {
        // Tiles of 1024 symbols staged in LDS as (I, Q) pairs, 8-way polyphase (sample c -> [c & 7][c >> 3]): thread g computes the
        // four symbols 4g..4g+3 of the tile from one window of MT + 6 samples; for a fixed window position all lanes read the same
        // phase at consecutive slots (conflict free), a tap is one broadcast read feeding 8 packed FMAs.
        constexpr int MM = MT ? MT : 1, TS = 1024, XPH = (2 * TS + MM - 1 + 7) / 8 + 1, NCH = (2 * TS + MM - 1 + 3) / 4;
        __shared__ float2 xt[8 * XPH];
        __shared__ float2 Wt[MM];
        if (tid < MM) Wt[tid] = make_float2(Ws[tid], -Ws[M + tid]);          // y = sum_k Wt[k] * x  (w = W0 - j W1)
        for (int n0 = 0; n0 < N; n0 += TS) {
            const int64_t sb = 2 * (int64_t)n0 - pad;                          // first sample of the tile window
            __syncthreads();
            for (int v = tid; v < NCH; v += 256) {
                const int64_t s4 = sb + 4 * v;
                float a4[4], b4[4];
                if (GEN) {                                                     // s4 is even: the chunk = noise words s4 / 2 and s4 / 2 + 1
                    float2 c[4];
                    if (s4 >= 0 && s4 + 3 < L) {
                        const f4u ua = *reinterpret_cast<const f4u *>(cs + s4), ub = *reinterpret_cast<const f4u *>(cs + s4 + 2);
                        c[0] = make_float2(ua.x, ua.y); c[1] = make_float2(ua.z, ua.w);
                        c[2] = make_float2(ub.x, ub.y); c[3] = make_float2(ub.z, ub.w);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++) c[i] = (s4 + i >= 0 && s4 + i < L) ? cs[s4 + i] : make_float2(0.f, 0.f);
                    }
                    if (s4 + 3 >= 0 && s4 < L) {
                        awgn_noise_pair((uint32_t)(s4 >> 1), run, vg.frame, 0, vg.seed, sigma, c[0], c[1]);
                        awgn_noise_pair((uint32_t)(s4 >> 1) + 1u, run, vg.frame, 0, vg.seed, sigma, c[2], c[3]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool ok = s4 + i >= 0 && s4 + i < L;             // outside the frame: the equalizer's zero padding, not noise
                        a4[i] = ok ? c[i].x : 0.f;
                        b4[i] = ok ? c[i].y : 0.f;
                    }
                } else if (s4 >= 0 && s4 + 3 < L) {                            // one 16-byte load per row (any 4-byte alignment)
                    const f4u ua = *reinterpret_cast<const f4u *>(x0 + s4), ub = *reinterpret_cast<const f4u *>(x1 + s4);
                    a4[0] = ua.x; a4[1] = ua.y; a4[2] = ua.z; a4[3] = ua.w;
                    b4[0] = ub.x; b4[1] = ub.y; b4[2] = ub.z; b4[3] = ub.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool ok = s4 + i >= 0 && s4 + i < L;
                        a4[i] = ok ? x0[s4 + i] : 0.f;
                        b4[i] = ok ? x1[s4 + i] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = 4 * v + i;
                    xt[(c & 7) * XPH + (c >> 3)] = make_float2(a4[i], b4[i]);
                }
            }
            __syncthreads();
            cacc acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = cacc0();
            constexpr int G8 = MM / 8;
#pragma unroll 1
            for (int g = 0; g < G8; g++) {                                     // taps 8g..8g+7: window positions 8g..8g+13
                const float2 *xg = xt + tid + g;
                float2 xw[14];
#pragma unroll
                for (int j = 0; j < 14; j++) xw[j] = xg[(j & 7) * XPH + (j >> 3)];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float2 w = Wt[8 * g + k];
#pragma unroll
                    for (int t = 0; t < 4; t++) cmac(acc[t], w.x, w.y, xw[2 * t + k]);
                }
            }
#pragma unroll
            for (int k = 8 * G8; k < MM; k++) {                                // remaining taps
                const float2 w = Wt[k];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int j = 2 * t + k;
                    cmac(acc[t], w.x, w.y, xt[(j & 7) * XPH + tid + (j >> 3)]);
                }
            }
            {
                const int nb = n0 + 4 * tid;
                float2 yv[4];
#pragma unroll
                for (int t = 0; t < 4; t++) yv[t] = cfin(acc[t]);
                if (nb + 3 < N) {                                              // 16-byte stores
                    *reinterpret_cast<f4u *>(y0 + nb) = f4u{yv[0].x, yv[1].x, yv[2].x, yv[3].x};
                    *reinterpret_cast<f4u *>(y1 + nb) = f4u{yv[0].y, yv[1].y, yv[2].y, yv[3].y};
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int n = nb + t;
                    if (n < N) {
                        if (nb + 3 >= N) { y0[n] = yv[t].x; y1[n] = yv[t].y; }
                        sa0 += fabsf(yv[t].x); sa1 += fabsf(yv[t].y);
                    }
                }
            }
        }
    } else {
        for (int n = tid; n < N; n += 256) {
            float yI = 0.f, yQ = 0.f;
            for (int k = 0; k < M; k++) {
                const int64_t sx = (int64_t)n * sps + k - pad;
                if (sx < 0 || sx >= L) continue;
                const float a_ = x0[sx], b_ = x1[sx];
                yI = fmaf(Ws[k], a_, yI); yI = fmaf(Ws[M + k], b_, yI);
                yQ = fmaf(Ws[k], b_, yQ); yQ = fmaf(-Ws[M + k], a_, yQ);
            }
            y0[n] = yI; y1[n] = yQ;
            sa0 += fabsf(yI); sa1 += fabsf(yQ);
        }
    }
    block_reduce3<256>(sa0, sa1, 0.f, red);               // (its barriers also order the y writes before the reads below)
    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = amp_g[i];
    const float A = amp_mean[run], ivar = 1.0f / var[run];
    const float s0 = A / (red[0] / (float)N), s1 = A / (red[1] / (float)N);
    const int NE = N < VAL_NE ? N : VAL_NE;
    // square-QAM levels are equidistant: the nearest level is a rounding (any other level set takes the search loop)
    const float a0 = amp[0], delta = amp[1] - amp[0], rdelta = 1.0f / delta;
    bool uniform = delta > 0.f;
#pragma unroll
    for (int i = 2; i < NLEV; i++) uniform = uniform && fabsf(amp[i] - (a0 + (float)i * delta)) <= 1e-6f * delta;
    for (int nb = 4 * tid; nb < N; nb += 4 * 256) {          // four consecutive symbols per thread: 16-byte loads of y
        float yI4[4], yQ4[4];
        if (nb + 3 < N) {
            const f4u ua = *reinterpret_cast<const f4u *>(y0 + nb), ub = *reinterpret_cast<const f4u *>(y1 + nb);
            yI4[0] = ua.x; yI4[1] = ua.y; yI4[2] = ua.z; yI4[3] = ua.w;
            yQ4[0] = ub.x; yQ4[1] = ub.y; yQ4[2] = ub.z; yQ4[3] = ub.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) { yI4[t] = nb + t < N ? y0[nb + t] : 0.f; yQ4[t] = nb + t < N ? y1[nb + t] : 0.f; }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int n = nb + t;
            if (n >= N) break;
            const float yI = yI4[t] * s0, yQ = yQ4[t] * s1;
            int dI = 0, dQ = 0;
            float bI = 3.0e38f, bQ = 3.0e38f;
            if (uniform) {
                dI = min(max((int)rintf((yI - a0) * rdelta), 0), NLEV - 1);
                dQ = min(max((int)rintf((yQ - a0) * rdelta), 0), NLEV - 1);
                const float dd = yI - (a0 + (float)dI * delta);                // only stabilises the softmax of E below
                bI = dd * dd;
            } else {
#pragma unroll
                for (int i = 0; i < NLEV; i++) {
                    const float eI = (yI - amp[i]) * (yI - amp[i]), eQ = (yQ - amp[i]) * (yQ - amp[i]);
                    if (eI < bI) { bI = eI; dI = i; }
                    if (eQ < bQ) { bQ = eQ; dQ = i; }
                }
            }
            decs[n] = (unsigned char)(dI | (dQ << 4));
            if (n < NE) {
                float ssum = 0.f, e1 = 0.f;
#pragma unroll
                for (int i = 0; i < NLEV; i++) {
                    const float d = yI - amp[i], w = __expf(bI * ivar - d * d * ivar);
                    ssum += w;
                    e1 = fmaf(amp[i], w, e1);
                }
                E[n] = e1 / ssum;
            }
        }
    }
    __syncthreads();
    validate_tail<256, NLEV>(N, n_shift, decs, E, NE, data + (size_t)run * 2 * N, red, corr, &sh_s, ser_out + run, shift_out ? shift_out + run : nullptr);
}

template <int NLEV, bool GEN = false>
static int launch_validate(int R, int N, int sps, int M, int n_shift, const float *x, const float *W, const float *amp, const float *amp_mean,
                           const float *var, const __half *data, float *yws, float *ser, int *shift, hipStream_t st, const ValGen &vg = ValGen{})
{
    const size_t lds = ((size_t)N + 15) & ~(size_t)15;
#define VAEQ_VAL(MM)                                                                                                         \
    {                                                                                                                        \
        auto k = awgn_validate_kernel<NLEV, MM, GEN && (MM > 0)>;                                                            \
        if (lds > 32 * 1024 &&                                                                                               \
            hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return VAEQ_ERR_LDS;                                                                                             \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, N, sps, M, n_shift, x, W, amp, amp_mean, var, data, yws, ser, shift, vg); \
    }
    if (sps == 2 && M == 25) VAEQ_VAL(25)
    else if (sps == 2 && M == 17) VAEQ_VAL(17)
    else if (sps == 2 && M == 9) VAEQ_VAL(9)
    else if (GEN) return VAEQ_ERR_SHAPE;                       // the noise-on-load form exists for the baked tap counts only
    else VAEQ_VAL(0)
#undef VAEQ_VAL
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}


// Stand-alone ELBO of the single-polarisation variants for a given q (values only):
//   func_VAELE_MQAM_shaping.loss_function (:63-95):  nm log C + sum q log(q / P + 1e-12)      (P != nullptr)
//   func_VAENN_MQAM.loss_function (:63-95):          nm log C + sum q log(q + 1e-12)          (P == nullptr)
// One workgroup per run; q[R][2n][B], x[R][2][B*sps], h[R][2][M] -> loss[R].
template <int NLEV>
__global__ __launch_bounds__(256) void awgn_loss_kernel(int B, int sps, int M, const float *__restrict__ q, const float *__restrict__ x,
                                                        const float *__restrict__ h, const float *__restrict__ amp_g, const float *__restrict__ P,
                                                        float *__restrict__ loss)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    __shared__ float red[64];
    __shared__ float hs[2 * 64], VS[64];
    const int run = blockIdx.x, tid = threadIdx.x;
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh;
    float *mu = sm, *vr = sm + 2 * B;
    float amp[NLEV], invP[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; invP[i] = P ? 1.0f / P[(size_t)run * NLEV + i] : 1.0f; }
    for (int i = tid; i < 2 * M; i += 256) hs[i] = h[(size_t)run * 2 * M + i];
    const float *qr = q + (size_t)run * 2 * NLEV * B, *x0 = x + (size_t)run * 2 * L, *x1 = x0 + L;
    float klsum = 0.f;
    for (int it = tid; it < 2 * B; it += 256) {
        const int c = it / B, n = it - c * B;
        const bool inr = (n >= mh) && (n < B - mh);
        float qq[NLEV], e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) { qq[i] = qr[(size_t)(c * NLEV + i) * B + n]; e1 = fmaf(amp[i], qq[i], e1); }
#pragma unroll
        for (int i = 0; i < NLEV; i++) {
            const float d = amp[i] - e1;
            e2 = fmaf(qq[i] * d, d, e2);
            if (inr) klsum = fmaf(qq[i], __logf(qq[i] * invP[i] + 1e-12f), klsum);
        }
        mu[it] = e1; vr[it] = e2;
    }
    __syncthreads();
    float se = 0.f;
    for (int t = tid; t < nm; t += 256) {
        float dr = 0.f, di = 0.f;
        for (int j = (t + Mh) % sps; j <= Mh; j += sps) {
            const int np = (t + Mh - j) / sps;
            const float a_ = mu[np], b_ = mu[B + np], c_ = hs[j], d_ = hs[M + j];
            dr = fmaf(c_, a_, dr); dr = fmaf(-d_, b_, dr);
            di = fmaf(c_, b_, di); di = fmaf(d_, a_, di);
        }
        const float er = x0[mh + t] - dr, ei = x1[mh + t] - di;
        se += er * er + ei * ei;
    }
    for (int j = tid; j < M; j += 256) {
        const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
        float acc = 0.f;
        for (int np = lo; np <= hi_; np++) acc += vr[np] + vr[B + np];
        VS[j] = acc;
    }
    block_reduce3<256>(se, klsum, 0.f, red);
    if (tid == 0) {
        float C = red[0];
        for (int j = 0; j < M; j++) C = fmaf(hs[j] * hs[j] + hs[M + j] * hs[M + j], VS[j], C);
        loss[run] = (float)nm * logf(C) + red[1];
    }
}


// Backward of the stand-alone AWGN ELBO (for the autograd wrappers; the fused training kernels do not use it):
// g_up[R] = upstream d/dloss -> gq[R][2n][B] = dL/dq, gh[R][2][M] = dL/dh.  P == nullptr: the VAE-NN form (entropy).
template <int NLEV>
__global__ __launch_bounds__(256) void awgn_loss_bwd_kernel(int B, int sps, int M, const float *__restrict__ q, const float *__restrict__ x,
                                                            const float *__restrict__ h, const float *__restrict__ amp_g,
                                                            const float *__restrict__ P, const float *__restrict__ g_up, float *__restrict__ gq,
                                                            float *__restrict__ gh)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    __shared__ float red[64];
    __shared__ float hs[2 * 64], VS[64];
    const int run = blockIdx.x, tid = threadIdx.x;
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh;
    float *mu = sm, *vr = sm + 2 * B, *es = sm + 4 * B;       // es[2][nm]
    float amp[NLEV], invP[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; invP[i] = P ? 1.0f / P[(size_t)run * NLEV + i] : 1.0f; }
    for (int i = tid; i < 2 * M; i += 256) hs[i] = h[(size_t)run * 2 * M + i];
    const float *qr = q + (size_t)run * 2 * NLEV * B, *x0 = x + (size_t)run * 2 * L, *x1 = x0 + L;
    for (int it = tid; it < 2 * B; it += 256) {
        const int c = it / B, n = it - c * B;
        float e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) e1 = fmaf(amp[i], qr[(size_t)(c * NLEV + i) * B + n], e1);
#pragma unroll
        for (int i = 0; i < NLEV; i++) { const float d = amp[i] - e1; e2 = fmaf(qr[(size_t)(c * NLEV + i) * B + n] * d, d, e2); }
        mu[it] = e1; vr[it] = e2;
    }
    __syncthreads();
    float se = 0.f;
    for (int t = tid; t < nm; t += 256) {
        float dr = 0.f, di = 0.f;
        for (int j = (t + Mh) % sps; j <= Mh; j += sps) {
            const int np = (t + Mh - j) / sps;
            const float a_ = mu[np], b_ = mu[B + np], c_ = hs[j], d_ = hs[M + j];
            dr = fmaf(c_, a_, dr); dr = fmaf(-d_, b_, dr);
            di = fmaf(c_, b_, di); di = fmaf(d_, a_, di);
        }
        const float er = x0[mh + t] - dr, ei = x1[mh + t] - di;
        es[t] = er; es[nm + t] = ei;
        se += er * er + ei * ei;
    }
    for (int j = tid; j < M; j += 256) {
        const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
        float acc = 0.f;
        for (int np = lo; np <= hi_; np++) acc += vr[np] + vr[B + np];
        VS[j] = acc;
    }
    block_reduce3<256>(se, 0.f, 0.f, red);
    float C = red[0];
    for (int j = 0; j < M; j++) C = fmaf(hs[j] * hs[j] + hs[M + j] * hs[M + j], VS[j], C);
    const float up = g_up[run], gC = up * (float)nm / C;
    for (int j = tid; j < M; j += 256) {
        const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
        float ar = 0.f, ai = 0.f;
        for (int np = lo; np <= hi_; np++) {
            const int t = np * sps - Mh + j;
            const float a_ = es[t], b_ = es[nm + t], c_ = mu[np], d_ = mu[B + np];
            ar = fmaf(a_, c_, ar); ar = fmaf(b_, d_, ar);
            ai = fmaf(b_, c_, ai); ai = fmaf(-a_, d_, ai);
        }
        gh[(size_t)run * 2 * M + j] = gC * (-2.0f * ar + 2.0f * hs[j] * VS[j]);
        gh[(size_t)run * 2 * M + M + j] = gC * (-2.0f * ai + 2.0f * hs[M + j] * VS[j]);
    }
    float *gqr = gq + (size_t)run * 2 * NLEV * B;
    for (int n = tid; n < B; n += 256) {
        const int sx = n * sps;
        const int jlo = max(0, Mh - sx), jhi = min(Mh, nm - 1 + Mh - sx);
        const float *er = es + (sx - Mh), *ei = er + nm;
        float pr = 0.f, pi = 0.f, ph = 0.f;
        for (int j = jlo; j <= jhi; j++) {
            const float a_ = er[j], b_ = ei[j], c_ = hs[j], d_ = hs[M + j];
            pr = fmaf(a_, c_, pr); pr = fmaf(b_, d_, pr);
            pi = fmaf(b_, c_, pi); pi = fmaf(-a_, d_, pi);
            ph = fmaf(c_, c_, ph); ph = fmaf(d_, d_, ph);
        }
        const float gv = gC * ph;
        const bool inr = (n >= mh) && (n < B - mh);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float gmu = -2.0f * gC * (c ? pi : pr) - 2.0f * mu[c * B + n] * gv;
#pragma unroll
            for (int i = 0; i < NLEV; i++) {
                const float qq = qr[(size_t)(c * NLEV + i) * B + n];
                float g = amp[i] * gmu + amp[i] * amp[i] * gv;
                if (inr) { const float r = qq * invP[i], re = r + 1e-12f; g += up * (__logf(re) + r / re); }
                gqr[(size_t)(c * NLEV + i) * B + n] = g;
            }
        }
    }
}

// Backward of twoFIR.forward (func_VAELE_MQAM_shaping.py:214-231): upstream gq[R][2n][N] (and optionally gy on the un-normalised
// output) -> gW[R][2][M].  Recomputes the forward (y, mean |y|, yhat, q), then softmax backward with dz_i/dyhat = -2 (yhat - a_i) / var,
// the normalisation's Jacobian and the tap correlation.  One workgroup per run, y and dL/dy in LDS.
template <int NLEV>
__global__ __launch_bounds__(256) void awgn_forward_bwd_kernel(int N, int sps, int M, const float *__restrict__ x, const float *__restrict__ W,
                                                               const float *__restrict__ amp_g, const float *__restrict__ amp_mean,
                                                               const float *__restrict__ var, const float *__restrict__ gq,
                                                               const float *__restrict__ gy_up, float *__restrict__ gW)
{
    extern __shared__ float4 smem4[];
    float *ys = reinterpret_cast<float *>(smem4), *gys = ys + 2 * N;
    __shared__ float Ws[2 * 64];
    __shared__ float red[64];
    const int run = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2 * M; i += 256) Ws[i] = W[(size_t)run * 2 * M + i];
    __syncthreads();
    const int L = N * sps, pad = (M - 1) / 2;
    const float *x0 = x + (size_t)run * 2 * L, *x1 = x0 + L;
    float sa0 = 0.f, sa1 = 0.f;
    for (int n = tid; n < N; n += 256) {
        float yI = 0.f, yQ = 0.f;
        for (int k = 0; k < M; k++) {
            const int sx = n * sps + k - pad;
            if (sx < 0 || sx >= L) continue;
            const float a_ = x0[sx], b_ = x1[sx];
            yI = fmaf(Ws[k], a_, yI); yI = fmaf(Ws[M + k], b_, yI);
            yQ = fmaf(Ws[k], b_, yQ); yQ = fmaf(-Ws[M + k], a_, yQ);
        }
        ys[n] = yI; ys[N + n] = yQ;
        sa0 += fabsf(yI); sa1 += fabsf(yQ);
    }
    block_reduce3<256>(sa0, sa1, 0.f, red);
    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = amp_g[i];
    const float A = amp_mean[run], ivar = 1.0f / var[run];
    const float m0 = red[0] / (float)N, m1 = red[1] / (float)N;
    __syncthreads();
    const float *gqr = gq + (size_t)run * 2 * NLEV * N;
    float dt0 = 0.f, dt1 = 0.f;
    for (int it = tid; it < 2 * N; it += 256) {
        const int c = it / N, n = it - c * N;
        const float yh = ys[it] / (c ? m1 : m0) * A;
        float z[NLEV], zmax = -3.0e38f, ssum = 0.f, dot = 0.f, g = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) { const float d = yh - amp[i]; z[i] = -(d * d * ivar); zmax = fmaxf(zmax, z[i]); }
#pragma unroll
        for (int i = 0; i < NLEV; i++) { z[i] = __expf(z[i] - zmax); ssum += z[i]; }
#pragma unroll
        for (int i = 0; i < NLEV; i++) { z[i] /= ssum; dot = fmaf(z[i], gqr[(size_t)(c * NLEV + i) * N + n], dot); }
#pragma unroll
        for (int i = 0; i < NLEV; i++) g = fmaf(z[i] * (gqr[(size_t)(c * NLEV + i) * N + n] - dot), -2.0f * (yh - amp[i]) * ivar, g);
        gys[it] = g;                                           // dL/dyhat
        if (c) dt1 = fmaf(g, ys[it], dt1); else dt0 = fmaf(g, ys[it], dt0);
    }
    block_reduce3<256>(dt0, dt1, 0.f, red);
    {
        const float s0_ = A / m0, s1_ = A / m1, k0_ = red[0] * A / (m0 * m0) / (float)N, k1_ = red[1] * A / (m1 * m1) / (float)N;
        __syncthreads();
        for (int n = tid; n < N; n += 256) {                   // normalisation backward (:228) + the upstream gradient on `out`
            const float yI = ys[n], yQ = ys[N + n];
            const float sgI = (float)(yI > 0.f) - (float)(yI < 0.f), sgQ = (float)(yQ > 0.f) - (float)(yQ < 0.f);
            gys[n] = gys[n] * s0_ - k0_ * sgI + (gy_up ? gy_up[(size_t)run * 2 * N + n] : 0.f);
            gys[N + n] = gys[N + n] * s1_ - k1_ * sgQ + (gy_up ? gy_up[(size_t)run * 2 * N + N + n] : 0.f);
        }
    }
    __syncthreads();
    for (int k = tid; k < M; k += 256) {                       // dL/dW0[k] = sum gI x0 + gQ x1, dL/dW1[k] = sum gI x1 - gQ x0
        float g0 = 0.f, g1 = 0.f;
        for (int n = 0; n < N; n++) {
            const int sx = n * sps + k - pad;
            if (sx < 0 || sx >= L) continue;
            const float a_ = gys[n], b_ = gys[N + n], c_ = x0[sx], d_ = x1[sx];
            g0 = fmaf(a_, c_, g0); g0 = fmaf(b_, d_, g0);
            g1 = fmaf(a_, d_, g1); g1 = fmaf(-b_, c_, g1);
        }
        gW[(size_t)run * 2 * M + k] = g0;
        gW[(size_t)run * 2 * M + M + k] = g1;
    }
}

template <int NT, int NLEV>
static int launch_awgn(const vaeq_awgn_args &a, size_t lds, hipStream_t st)
{
    auto k = awgn_train_kernel<NT, NLEV>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    note_kernel("vaeq::awgn_train_kernel<%d, %d>", NT, NLEV);
    hipLaunchKernelGGL(k, dim3(a.R), dim3(NT), lds, st, a);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

template <int NT>
static int launch_awgn_lev(const vaeq_awgn_args &a, size_t lds, hipStream_t st)
{
    switch (a.n_lev) {
    case 2: return launch_awgn<NT, 2>(a, lds, st);
    case 4: return launch_awgn<NT, 4>(a, lds, st);
    case 8: return launch_awgn<NT, 8>(a, lds, st);
    }
    return VAEQ_ERR_SHAPE;
}

}  // namespace vaeq

extern "C" int64_t vaeq_awgn_lds_bytes(int32_t B, int32_t sps, int32_t M, int32_t n_lev)
{
    if (B <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || !(n_lev == 2 || n_lev == 4 || n_lev == 8)) return VAEQ_ERR_SHAPE;
    if ((int64_t)B * sps - 2 * (M / 2) <= 0 || B <= 2 * (M / 2)) return VAEQ_ERR_SHAPE;
    return (int64_t)vaeq::awgn_layout(B, sps, M).total * 4;
}

extern "C" int vaeq_awgn_train(const vaeq_awgn_args *pa, void *stream)
{
    if (!pa) return VAEQ_ERR_NULL;
    const vaeq_awgn_args &a = *pa;
    if (a.R == 0) return VAEQ_OK;                              // an empty batch owns no memory: its pointers may be NULL
    if (!a.rx || !a.W || !a.h || !a.adam_mW || !a.adam_vW || !a.adam_xW || !a.adam_mh || !a.adam_vh || !a.adam_xh || !a.step ||
        !a.amp || !a.P || !a.amp_mean || !a.var || !a.lr)
        return VAEQ_ERR_NULL;
    const int64_t lds = vaeq_awgn_lds_bytes(a.B, a.sps, a.M, a.n_lev);
    if (lds < 0) return (int)lds;
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    if (a.R < 0 || a.steps <= 0) return VAEQ_ERR_SHAPE;
    if ((int64_t)a.steps * a.B * a.sps > a.S) return VAEQ_ERR_SHAPE;
    if (a.R == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a.threads == 1 && !vaeq::awgn_wave_supported(a)) return VAEQ_ERR_SHAPE;
    if ((a.threads == 0 || a.threads == 1) && vaeq::awgn_wave_supported(a)) return vaeq::launch_awgn_wave(a, st);
    switch (a.threads) {
    case 0:
    case 256: return vaeq::launch_awgn_lev<256>(a, (size_t)lds, st);
    case 128: return vaeq::launch_awgn_lev<128>(a, (size_t)lds, st);
    case 64: return vaeq::launch_awgn_lev<64>(a, (size_t)lds, st);
    }
    return VAEQ_ERR_SHAPE;
}

extern "C" int vaeq_awgn_forward(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *W,
                                 const float *amp, const float *amp_mean, const float *var, float *q, float *y, void *stream)
{
    if (R == 0 || N == 0) return VAEQ_OK;                      // an empty batch owns no memory: its pointers may be NULL
    if (!x || !W || !amp || !amp_mean || !var || !y) return VAEQ_ERR_NULL;
    if (R < 0 || N < 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63) return VAEQ_ERR_SHAPE;
    if (R == 0 || N == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (n_lev) {
    case 2: hipLaunchKernelGGL(vaeq::awgn_forward_kernel<2>, dim3(R), dim3(256), 0, st, N, sps, M, x, W, amp, amp_mean, var, q, y); break;
    case 4: hipLaunchKernelGGL(vaeq::awgn_forward_kernel<4>, dim3(R), dim3(256), 0, st, N, sps, M, x, W, amp, amp_mean, var, q, y); break;
    case 8: hipLaunchKernelGGL(vaeq::awgn_forward_kernel<8>, dim3(R), dim3(256), 0, st, N, sps, M, x, W, amp, amp_mean, var, q, y); break;
    default: return VAEQ_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_awgn_validate(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t n_shift, const float *x, const float *W,
                                  const float *amp, const float *amp_mean, const float *var, const void *data_f16, float *y_ws, float *ser,
                                  int32_t *shift, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!x || !W || !amp || !amp_mean || !var || !data_f16 || !y_ws || !ser) return VAEQ_ERR_NULL;
    if (R < 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || n_shift <= 0 || n_shift > vaeq::VAL_MAXSHIFT) return VAEQ_ERR_SHAPE;
    if (N < 64 || N > 65536) return VAEQ_ERR_SHAPE;           // decisions live in LDS (N bytes); 22 + n_shift symbols are trimmed
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __half *d = reinterpret_cast<const __half *>(data_f16);
    switch (n_lev) {
    case 2: return vaeq::launch_validate<2>(R, (int)N, sps, M, n_shift, x, W, amp, amp_mean, var, d, y_ws, ser, shift, st);
    case 4: return vaeq::launch_validate<4>(R, (int)N, sps, M, n_shift, x, W, amp, amp_mean, var, d, y_ws, ser, shift, st);
    case 8: return vaeq::launch_validate<8>(R, (int)N, sps, M, n_shift, x, W, amp, amp_mean, var, d, y_ws, ser, shift, st);
    }
    return VAEQ_ERR_SHAPE;
}

// vaeq_awgn_validate on x = clean frame + noise made while staging (include/vaeq.h)
extern "C" int vaeq_awgn_validate_gen(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t n_shift, const float *sig, int32_t Ls,
                                      const float *power_ws, const float *snr_db, const float *sigma_fixed, uint64_t seed, uint32_t frame,
                                      const float *W, const float *amp, const float *amp_mean, const float *var, const void *data_f16,
                                      float *y_ws, float *ser, int32_t *shift, float *sigma_out, void *stream)
{
    if (R == 0) return VAEQ_OK;
    if (!sig || !power_ws || (!snr_db && !sigma_fixed) || !W || !amp || !amp_mean || !var || !data_f16 || !y_ws || !ser) return VAEQ_ERR_NULL;
    if (R < 0 || sps != 2 || (M != 9 && M != 17 && M != 25) || n_shift <= 0 || n_shift > vaeq::VAL_MAXSHIFT) return VAEQ_ERR_SHAPE;
    if (N < 64 || N > 65536 || Ls < sps * N) return VAEQ_ERR_SHAPE;      // the clean frame holds at least the sps * N samples vaeq_gen_awgn keeps
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __half *d = reinterpret_cast<const __half *>(data_f16);
    vaeq::ValGen vg{reinterpret_cast<const float2 *>(sig), power_ws, snr_db, sigma_fixed, sigma_out, seed, frame, Ls, (Ls + 2047) / 2048};
    switch (n_lev) {
    case 2: return vaeq::launch_validate<2, true>(R, (int)N, sps, M, n_shift, nullptr, W, amp, amp_mean, var, d, y_ws, ser, shift, st, vg);
    case 4: return vaeq::launch_validate<4, true>(R, (int)N, sps, M, n_shift, nullptr, W, amp, amp_mean, var, d, y_ws, ser, shift, st, vg);
    case 8: return vaeq::launch_validate<8, true>(R, (int)N, sps, M, n_shift, nullptr, W, amp, amp_mean, var, d, y_ws, ser, shift, st, vg);
    }
    return VAEQ_ERR_SHAPE;
}

extern "C" int vaeq_awgn_loss(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x, const float *h,
                              const float *amp, const float *P, float *loss, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!q || !x || !h || !amp || !loss) return VAEQ_ERR_NULL;
    if (R < 0 || B <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || (int64_t)B * sps - 2 * (M / 2) <= 0 || B <= 2 * (M / 2)) return VAEQ_ERR_SHAPE;
    const size_t lds = (size_t)4 * B * sizeof(float);
    if (lds > 150 * 1024) return VAEQ_ERR_LDS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VAEQ_AL(NL)                                                                                                                     \
    {                                                                                                                                   \
        auto k = vaeq::awgn_loss_kernel<NL>;                                                                                            \
        if (lds > 32 * 1024 &&                                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
            return VAEQ_ERR_LDS;                                                                                                        \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, B, sps, M, q, x, h, amp, P, loss);                                           \
    }
    switch (n_lev) {
    case 2: VAEQ_AL(2) break;
    case 4: VAEQ_AL(4) break;
    case 8: VAEQ_AL(8) break;
    default: return VAEQ_ERR_SHAPE;
    }
#undef VAEQ_AL
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_awgn_loss_bwd(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x, const float *h,
                                  const float *amp, const float *P, const float *g_up, float *gq, float *gh, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!q || !x || !h || !amp || !g_up || !gq || !gh) return VAEQ_ERR_NULL;
    if (R < 0 || B <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || (int64_t)B * sps - 2 * (M / 2) <= 0 || B <= 2 * (M / 2)) return VAEQ_ERR_SHAPE;
    const size_t lds = ((size_t)4 * B + 2 * ((size_t)B * sps - 2 * (M / 2))) * sizeof(float);
    if (lds > 150 * 1024) return VAEQ_ERR_LDS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VAEQ_ALB(NL)                                                                                                                    \
    {                                                                                                                                   \
        auto k = vaeq::awgn_loss_bwd_kernel<NL>;                                                                                        \
        if (lds > 32 * 1024 &&                                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
            return VAEQ_ERR_LDS;                                                                                                        \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, B, sps, M, q, x, h, amp, P, g_up, gq, gh);                                   \
    }
    switch (n_lev) {
    case 2: VAEQ_ALB(2) break;
    case 4: VAEQ_ALB(4) break;
    case 8: VAEQ_ALB(8) break;
    default: return VAEQ_ERR_SHAPE;
    }
#undef VAEQ_ALB
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_awgn_forward_bwd(int32_t R, int32_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *W, const float *amp,
                                     const float *amp_mean, const float *var, const float *gq, const float *gy, float *gW, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!x || !W || !amp || !amp_mean || !var || !gq || !gW) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63) return VAEQ_ERR_SHAPE;
    const size_t lds = (size_t)4 * N * sizeof(float);
    if (lds > 150 * 1024) return VAEQ_ERR_LDS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VAEQ_AFB(NL)                                                                                                                    \
    {                                                                                                                                   \
        auto k = vaeq::awgn_forward_bwd_kernel<NL>;                                                                                     \
        if (lds > 32 * 1024 &&                                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
            return VAEQ_ERR_LDS;                                                                                                        \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, N, sps, M, x, W, amp, amp_mean, var, gq, gy, gW);                            \
    }
    switch (n_lev) {
    case 2: VAEQ_AFB(2) break;
    case 4: VAEQ_AFB(4) break;
    case 8: VAEQ_AFB(8) break;
    default: return VAEQ_ERR_SHAPE;
    }
#undef VAEQ_AFB
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}
