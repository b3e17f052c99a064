// vaeq_dp.hip -- dual-polarisation VAE-LE / VAEflex training loop for gfx950 (MI355X).
//
// One workgroup = one independent run.  The whole minibatch step
//   butterfly FIR -> per-axis soft demapper -> ELBO -> analytic backward -> Adam
// runs out of LDS/registers; HBM sees only the received-sample stream (read once,
// coalesced) and the q / y / loss outputs (written once, coalesced along the symbol
// axis).  Steps of a run are strictly sequential (Adam state carries over), so the
// grid is the sweep: runs x 1.
//
// Math (reference = optical_DP_channel/shared_funcs.py):
//   FIR     y[o,n]   = sum_p sum_k w[o,p,k] x[p, n*sps + k - M/2],  w = W[o,p] + j W[o,2+p]      (:500-518)
//   demap   q_i      = softmax_i( -(y_c - a_i)^2 / (2 var[o]) - nu_sc a_i^2 )                     (:521-523)
//   moments mu = E_q[a], v = Var_q[a]                                                            (:107-113)
//   ELBO    D[chi,t] = sum_nu sum_j h[chi,nu,j] U[nu, t+Mh-j],  U = zero-stuffed mu              (:123-127)
//           C[chi]   = sum_t |x[chi,mh+t] - D|^2 + sum_{nu,j} |h|^2 VS[nu,j]                     (:128-134)
//           loss     = nm sum_chi log C[chi] + sum q log(q/P + 1e-12)                            (:131-136)
// Backward, closed form.  With gC = nm/C, e = x - D:
//   dL/dh[chi,nu,j] = gC[chi] ( -2 sum_t e[chi,t] conj(U[nu,t+Mh-j]) + 2 h VS[nu,j] )
//   dL/dU[nu,s]     = -2 sum_chi gC[chi] sum_j e[chi,s-Mh+j] conj(h[chi,nu,j])
//   G_V[nu,s]       = sum_chi gC[chi] sum_j |h[chi,nu,j]|^2 [0 <= s-Mh+j < nm]
// and the whole  q -> softmax -> y  chain collapses onto three per-symbol moments of q
// (derivation in DESIGN.md):
//   dL/dy_c[o,n] = ( dU_c * v + G_V * T3 + Kc ) / var[o]
//     v  = sum q_i (a_i-mu)^2,  T3 = sum q_i (a_i-mu)^3,
//     Kc = sum q_i (a_i-mu) k_i,  k_i = [mh<=n<B-mh] ( log(q_i/P_i+eps) + (q_i/P_i)/(q_i/P_i+eps) )
//   dL/dw[o,p,k] = sum_n dL/dy[o,n] conj(x[p, n*sps + k - M/2])
// so q never has to be kept for the backward pass: it is written to HBM once and forgotten.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vaeq.h"
#include "vaeq_common.h"

namespace vaeq {

// wave-per-run fast path (vaeq_dp_wave.hip)
bool dp_wave_supported(const vaeq_dp_args &a);
int launch_dp_wave(const vaeq_dp_args &a, hipStream_t st);

// ---------------------------------------------------------------- LDS carve
struct DPLayout {
    int L, mh, Mh, nm, Lp;
    int xs, Ws, hs, mW, vW, mH, vH, gW, gH, mu, vr, t3, kc, gy, es, VS, red, total;  // float offsets
};

__host__ __device__ inline int pad4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline DPLayout dp_layout(int B, int sps, int M)
{
    DPLayout l;
    l.L = B * sps;
    l.mh = M / 2;
    l.Mh = 2 * l.mh;
    l.nm = l.L - l.Mh;
    l.Lp = pad4(l.L + 2 * l.mh);
    int o = 0;
    auto take = [&](int n) { int r = o; o += pad4(n); return r; };
    l.xs = take(4 * l.Lp);
    l.Ws = take(8 * M);
    l.hs = take(8 * M);
    l.mW = take(8 * M);
    l.vW = take(8 * M);
    l.mH = take(8 * M);
    l.vH = take(8 * M);
    l.gW = take(8 * M);
    l.gH = take(8 * M);
    l.mu = take(4 * B);
    l.vr = take(4 * B);
    l.t3 = take(4 * B);
    l.kc = take(4 * B);
    l.gy = take(4 * B);
    l.es = take(4 * l.nm);
    l.VS = take(2 * M);
    l.red = take(64);
    l.total = o;
    return l;
}

// ---------------------------------------------------------------- kernel
template <int NT, int NLEV>
__global__ __launch_bounds__(NT) void dp_train_kernel(const vaeq_dp_args a)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    const int tid = threadIdx.x;
    const int run = blockIdx.x;
    const int B = a.B, sps = a.sps, M = a.M;
    const DPLayout l = dp_layout(B, sps, M);
    const int L = l.L, mh = l.mh, Mh = l.Mh, nm = l.nm, Lp = l.Lp;
    float *xs = sm + l.xs, *Ws = sm + l.Ws, *hs = sm + l.hs;
    float *mWs = sm + l.mW, *vWs = sm + l.vW, *mHs = sm + l.mH, *vHs = sm + l.vH, *gWs = sm + l.gW, *gHs = sm + l.gH;
    float *mu = sm + l.mu, *vr = sm + l.vr, *t3 = sm + l.t3, *kc = sm + l.kc, *gy = sm + l.gy, *es = sm + l.es;
    float *VS = sm + l.VS, *red = sm + l.red;
    const int NP = 8 * M;  // parameters per group

    // ---- per-run constants (uniform: scalar loads)
    float amp[NLEV], amp2[NLEV], invP[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) {
        amp[i] = a.amp[i];
        amp2[i] = amp[i] * amp[i];
        invP[i] = 1.0f / a.P[(size_t)run * NLEV + i];
    }
    const float var0 = a.var[run * 2 + 0], var1 = a.var[run * 2 + 1];
    const float nusc = a.nu_sc[run];
    const double lrW = (double)a.lr_W[run], lrH = (double)a.lr_h[run];

    // ---- state in
    for (int i = tid; i < NP; i += NT) {
        const size_t g = (size_t)run * NP + i;
        Ws[i] = a.W[g];
        hs[i] = a.h[g];
        mWs[i] = a.adam_mW[g];
        vWs[i] = a.adam_vW[g];
        mHs[i] = a.adam_mh[g];
        vHs[i] = a.adam_vh[g];
    }
    int step = a.step[run];
    __shared__ double BT[2][16];                               // beta^t at the start of the launch's first 16 frames (as in vaeq_dp_wave_kernel.h)
    if (tid < 16 && tid < a.n_frames) {
        const double t = (double)(step + tid * a.steps);
        BT[0][tid] = pow(0.9, t);
        BT[1][tid] = pow(0.999, t);
    }
    __syncthreads();
    double b1t = BT[0][0], b2t = BT[1][0];                     // carried in double, a running product inside a frame

    const int klen = a.keep_len, k0 = a.keep_off;
    const size_t No = (size_t)a.steps * klen;

    for (int f = 0; f < a.n_frames; f++) {
        if (f && f < 16) {                                     // restart beta^t like a launch does: frames per launch do not change the results
            b1t = BT[0][f];
            b2t = BT[1][f];
        }
        const float *rxf = a.rx + ((size_t)run * a.n_frames + f) * 4 * (size_t)a.S;
        float *qf = a.q_out ? a.q_out + ((size_t)run * a.n_frames + f) * (4 * NLEV) * No : nullptr;
        float *yf = a.y_out ? a.y_out + ((size_t)run * a.n_frames + f) * 4 * No : nullptr;
        float *ef = a.eq_out ? a.eq_out + ((size_t)run * a.n_frames + f) * 2 * No : nullptr;
        int8_t *df = a.dec_out ? a.dec_out + ((size_t)run * a.n_frames + f) * 4 * No : nullptr;
        for (int s = 0; s < a.steps; s++) {
            // ============ P0: window -> LDS, zero halo of mh samples (Conv1d padding, :494)
            const size_t s0 = (size_t)s * a.stride_sym * sps;
            for (int i = tid; i < 4 * Lp; i += NT) {
                const int row = i / Lp, c = i - row * Lp, sx = c - mh;
                xs[i] = (sx >= 0 && sx < L) ? rxf[(size_t)row * a.S + s0 + sx] : 0.0f;
            }
            __syncthreads();

            // ============ P1: FIR + demap + moments; item = (o, n)
            float klsum = 0.0f;
            for (int it = tid; it < 2 * B; it += NT) {
                const int o = it / B, n = it - o * B;
                float yI = 0.0f, yQ = 0.0f;
                const float *w = Ws + o * 4 * M;
#pragma unroll 1
                for (int p = 0; p < 2; p++) {
                    const float *xr = xs + (p * 2 + 0) * Lp + n * sps, *xi = xs + (p * 2 + 1) * Lp + n * sps;
                    const float *wr = w + p * M, *wi = w + (2 + p) * M;
                    for (int k = 0; k < M; k++) {
                        const float a_ = xr[k], b_ = xi[k], c_ = wr[k], d_ = wi[k];
                        yI = fmaf(c_, a_, yI);
                        yI = fmaf(-d_, b_, yI);
                        yQ = fmaf(c_, b_, yQ);
                        yQ = fmaf(d_, a_, yQ);
                    }
                }
                const bool kept = (n >= k0) && (n < k0 + klen);
                const size_t col = (size_t)s * klen + (n - k0);
                if (yf && kept) {
                    yf[(size_t)(o * 2 + 0) * No + col] = yI;
                    yf[(size_t)(o * 2 + 1) * No + col] = yQ;
                }
                const float varo = o ? var1 : var0;
                const float i2v = 0.5f / varo;
                const bool inr = (n >= mh) && (n < B - mh);  // KL slice, symbol index (:132)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float y = c ? yQ : yI;
                    float z[NLEV], zmax = -3.0e38f;
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        const float d = y - amp[i];
                        z[i] = -(d * d * i2v + nusc * amp2[i]);
                        zmax = fmaxf(zmax, z[i]);
                    }
                    float ssum = 0.0f;
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        z[i] = __expf(z[i] - zmax);
                        ssum += z[i];
                    }
                    const float rs = 1.0f / ssum;
                    float m1 = 0.0f;
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        z[i] *= rs;  // q_i
                        m1 = fmaf(amp[i], z[i], m1);
                    }
                    if (qf && kept) {
#pragma unroll
                        for (int i = 0; i < NLEV; i++) qf[(size_t)(o * 2 * NLEV + c * NLEV + i) * No + col] = z[i];
                    }
                    if (kept) {                                 // compact stand-ins for q in the epilogue (same values it would derive)
                        if (ef && c == 0) ef[(size_t)o * No + col] = m1;
                        if (df) {
                            float best = z[0];
                            int bi = 0;
#pragma unroll
                            for (int i = 1; i < NLEV; i++)
                                if (z[i] > best) { best = z[i]; bi = i; }
                            df[(size_t)(o * 2 + c) * No + col] = (int8_t)bi;
                        }
                    }
                    float m2 = 0.0f, m3 = 0.0f, kk = 0.0f;
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        const float d = amp[i] - m1, qd = z[i] * d;
                        m2 = fmaf(qd, d, m2);
                        m3 = fmaf(qd * d, d, m3);
                        if (inr) {
                            const float r = z[i] * invP[i], re = r + 1e-12f;
                            const float lg = __logf(re);
                            klsum = fmaf(z[i], lg, klsum);
                            kk = fmaf(qd, lg + r / re, kk);
                        }
                    }
                    const int ix = (o * 2 + c) * B + n;
                    mu[ix] = m1;
                    vr[ix] = m2;
                    t3[ix] = m3;
                    kc[ix] = kk;
                }
            }
            __syncthreads();

            // ============ P2: ELBO forward.  e = x - D (item = (chi,t)),  VS (item = (nu,j))
            float se0 = 0.0f, se1 = 0.0f;
            for (int it = tid; it < 2 * nm; it += NT) {
                const int chi = it / nm, t = it - chi * nm;
                float dr = 0.0f, di = 0.0f;
                const int j0 = (t + Mh) % sps;
#pragma unroll 1
                for (int v = 0; v < 2; v++) {
                    const float *hr = hs + ((chi * 2 + v) * 2 + 0) * M, *hi = hr + M;
                    const float *ur = mu + (v * 2 + 0) * B, *ui = mu + (v * 2 + 1) * B;
                    for (int j = j0; j <= Mh; j += sps) {
                        const int np = (t + Mh - j) / sps;
                        const float a_ = ur[np], b_ = ui[np], c_ = hr[j], d_ = hi[j];
                        dr = fmaf(c_, a_, dr);
                        dr = fmaf(-d_, b_, dr);
                        di = fmaf(d_, a_, di);
                        di = fmaf(c_, b_, di);
                    }
                }
                const float er = xs[(chi * 2 + 0) * Lp + Mh + t] - dr;  // x[chi, mh+t], halo offset mh
                const float ei = xs[(chi * 2 + 1) * Lp + Mh + t] - di;
                es[(chi * 2 + 0) * nm + t] = er;
                es[(chi * 2 + 1) * nm + t] = ei;
                const float e2 = er * er + ei * ei;
                if (chi) se1 += e2; else se0 += e2;
            }
            for (int it = tid; it < 2 * M; it += NT) {
                const int v = it / M, j = it - v * M;
                const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
                float acc = 0.0f;
                for (int np = lo; np <= hi_; np++) acc += vr[(v * 2 + 0) * B + np] + vr[(v * 2 + 1) * B + np];
                VS[it] = acc;
            }
            // block sums of se0, se1, klsum
            block_reduce3<NT>(se0, se1, klsum, red);  // ends with a barrier; results broadcast in red[0..2]

            // ============ P3: C, loss, gC (every thread redundantly from LDS: 4M-term sum, fixed order)
            float C0 = red[0], C1 = red[1];
            {
                float e0 = 0.0f, e1 = 0.0f;
                for (int i = 0; i < 2 * M; i++) {
                    const int v = i / M, j = i - v * M;
                    const float h0r = hs[((0 * 2 + v) * 2 + 0) * M + j], h0i = hs[((0 * 2 + v) * 2 + 1) * M + j];
                    const float h1r = hs[((1 * 2 + v) * 2 + 0) * M + j], h1i = hs[((1 * 2 + v) * 2 + 1) * M + j];
                    e0 = fmaf(h0r * h0r + h0i * h0i, VS[i], e0);
                    e1 = fmaf(h1r * h1r + h1i * h1i, VS[i], e1);
                }
                C0 += e0;
                C1 += e1;
            }
            const float gC0 = (float)nm / C0, gC1 = (float)nm / C1;
            if (tid == 0) {
                const size_t li = ((size_t)run * a.n_frames + f) * a.steps + s;
                if (a.loss) a.loss[li] = (float)nm * (logf(C0) + logf(C1)) + red[2];
                if (a.var_est) {
                    const size_t vi = ((size_t)run * a.n_frames + f) * 2 * a.steps + s;
                    a.var_est[vi] = C0 / (float)nm;
                    a.var_est[vi + a.steps] = C1 / (float)nm;
                }
            }

            // ============ P4a: dL/dh, item = (chi,nu,j) complex
            for (int it = tid; it < 4 * M; it += NT) {
                const int cv = it / M, j = it - cv * M, chi = cv >> 1, v = cv & 1;
                const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
                const float *er = es + (chi * 2 + 0) * nm, *ei = er + nm;
                const float *ur = mu + (v * 2 + 0) * B, *ui = mu + (v * 2 + 1) * B;
                float ar = 0.0f, ai = 0.0f;
                for (int np = lo; np <= hi_; np++) {
                    const int t = np * sps - Mh + j;
                    const float a_ = er[t], b_ = ei[t], c_ = ur[np], d_ = ui[np];
                    ar = fmaf(a_, c_, ar);
                    ar = fmaf(b_, d_, ar);
                    ai = fmaf(b_, c_, ai);
                    ai = fmaf(-a_, d_, ai);
                }
                const float gC = chi ? gC1 : gC0, vs = VS[v * M + j];
                const int ir = (cv * 2 + 0) * M + j, ii = ir + M;
                gHs[ir] = gC * (-2.0f * ar + 2.0f * hs[ir] * vs);
                gHs[ii] = gC * (-2.0f * ai + 2.0f * hs[ii] * vs);
            }
            // ============ P4b: dL/dU, G_V -> dL/dy, item = (nu, n)
            for (int it = tid; it < 2 * B; it += NT) {
                const int v = it / B, n = it - v * B, sx = n * sps;
                const int jlo = max(0, Mh - sx), jhi = min(Mh, nm - 1 + Mh - sx);
                float ur = 0.0f, ui = 0.0f, gv = 0.0f;
#pragma unroll 1
                for (int chi = 0; chi < 2; chi++) {
                    const float *er = es + (chi * 2 + 0) * nm + (sx - Mh), *ei = er + nm;
                    const float *hr = hs + ((chi * 2 + v) * 2 + 0) * M, *hi = hr + M;
                    float pr = 0.0f, pi = 0.0f, ph = 0.0f;
                    for (int j = jlo; j <= jhi; j++) {
                        const float a_ = er[j], b_ = ei[j], c_ = hr[j], d_ = hi[j];
                        pr = fmaf(a_, c_, pr);
                        pr = fmaf(b_, d_, pr);
                        pi = fmaf(b_, c_, pi);
                        pi = fmaf(-a_, d_, pi);
                        ph = fmaf(c_, c_, ph);
                        ph = fmaf(d_, d_, ph);
                    }
                    const float gC = chi ? gC1 : gC0;
                    ur = fmaf(-2.0f * gC, pr, ur);
                    ui = fmaf(-2.0f * gC, pi, ui);
                    gv = fmaf(gC, ph, gv);
                }
                const float iv = 1.0f / (v ? var1 : var0);
                const int iI = (v * 2 + 0) * B + n, iQ = iI + B;
                gy[iI] = iv * (ur * vr[iI] + gv * t3[iI] + kc[iI]);
                gy[iQ] = iv * (ui * vr[iQ] + gv * t3[iQ] + kc[iQ]);
            }
            __syncthreads();

            // ============ P5: dL/dw, item = (o,p,k) complex
            for (int it = tid; it < 4 * M; it += NT) {
                const int op = it / M, k = it - op * M, o = op >> 1, p = op & 1;
                const float *gI = gy + (o * 2 + 0) * B, *gQ = gI + B;
                const float *xr = xs + (p * 2 + 0) * Lp + k, *xi = xs + (p * 2 + 1) * Lp + k;
                float ar = 0.0f, ai = 0.0f;
                for (int n = 0; n < B; n++) {
                    const float a_ = gI[n], b_ = gQ[n], c_ = xr[n * sps], d_ = xi[n * sps];
                    ar = fmaf(a_, c_, ar);
                    ar = fmaf(b_, d_, ar);
                    ai = fmaf(b_, c_, ai);
                    ai = fmaf(-a_, d_, ai);
                }
                gWs[(o * 4 + p) * M + k] = ar;
                gWs[(o * 4 + 2 + p) * M + k] = ai;
            }
            __syncthreads();

            // ============ P6: Adam on both groups (torch.optim.Adam single-tensor path)
            step += 1;
            b1t *= 0.9;
            b2t *= 0.999;
            if (!a.no_update) {
                const double bc1 = 1.0 - b1t, bc2 = 1.0 - b2t;
                const float bc2s = (float)sqrt(bc2);
                const float ssW = (float)(lrW / bc1), ssH = (float)(lrH / bc1);
                for (int i = tid; i < 2 * NP; i += NT) {
                    const bool isW = i < NP;
                    const int ix = isW ? i : i - NP;
                    float *pp = isW ? Ws : hs, *mm = isW ? mWs : mHs, *vv = isW ? vWs : vHs;
                    const float g = isW ? gWs[ix] : gHs[ix];
                    adam_update(pp[ix], mm[ix], vv[ix], g, isW ? ssW : ssH, bc2s);
                }
            }
            __syncthreads();
        }
    }

    // ---- state out
    for (int i = tid; i < NP; i += NT) {
        const size_t g = (size_t)run * NP + i;
        if (!a.no_update) {
            a.W[g] = Ws[i];
            a.h[g] = hs[i];
            a.adam_mW[g] = mWs[i];
            a.adam_vW[g] = vWs[i];
            a.adam_mh[g] = mHs[i];
            a.adam_vh[g] = vHs[i];
        }
        if (a.dbg_gW) a.dbg_gW[g] = gWs[i];
        if (a.dbg_gh) a.dbg_gh[g] = gHs[i];
    }
    if (tid == 0 && !a.no_update) a.step[run] = step;
}

template <int NT, int NLEV>
static int launch_dp(const vaeq_dp_args &a, size_t lds, hipStream_t st)
{
    auto k = dp_train_kernel<NT, NLEV>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VAEQ_ERR_LDS;
    }
    note_kernel("vaeq::dp_train_kernel<%d, %d>", NT, NLEV);
    hipLaunchKernelGGL(k, dim3(a.R), dim3(NT), lds, st, a);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

template <int NT>
static int launch_dp_lev(const vaeq_dp_args &a, size_t lds, hipStream_t st)
{
    switch (a.n_lev) {
    case 2: return launch_dp<NT, 2>(a, lds, st);
    case 4: return launch_dp<NT, 4>(a, lds, st);
    case 8: return launch_dp<NT, 8>(a, lds, st);
    }
    return VAEQ_ERR_SHAPE;
}

}  // namespace vaeq

extern "C" int64_t vaeq_dp_lds_bytes(int32_t B, int32_t sps, int32_t M, int32_t n_lev)
{
    if (B <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || !(n_lev == 2 || n_lev == 4 || n_lev == 8)) return VAEQ_ERR_SHAPE;
    if ((int64_t)B * sps - 2 * (M / 2) <= 0) return VAEQ_ERR_SHAPE;  // needs nm > 0 residual samples; the KL slice mh <= n < B - mh may be empty (B <= 2 mh:
                                                                     // the reference's short batch_len options, Eval_run_DP.py:38 -- torch sums an empty slice to 0, shared_funcs.py:131-132)
    return (int64_t)vaeq::dp_layout(B, sps, M).total * 4;
}

extern "C" int vaeq_dp_train(const vaeq_dp_args *pa, void *stream)
{
    if (!pa) return VAEQ_ERR_NULL;
    const vaeq_dp_args &a = *pa;
    if (a.R == 0) return VAEQ_OK;                              // an empty batch owns no memory: its pointers may be NULL
    if (!a.rx || !a.W || !a.h || !a.adam_mW || !a.adam_vW || !a.adam_mh || !a.adam_vh || !a.step || !a.amp || !a.P ||
        !a.var || !a.nu_sc || !a.lr_W || !a.lr_h)
        return VAEQ_ERR_NULL;
    const int64_t lds = vaeq_dp_lds_bytes(a.B, a.sps, a.M, a.n_lev);
    if (lds < 0) return (int)lds;
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    if (a.R < 0 || a.n_frames <= 0 || a.steps <= 0 || a.stride_sym <= 0) return VAEQ_ERR_SHAPE;
    if (a.keep_off < 0 || a.keep_len <= 0 || a.keep_off + a.keep_len > a.B) return VAEQ_ERR_SHAPE;
    if (((int64_t)(a.steps - 1) * a.stride_sym + a.B) * a.sps > a.S) return VAEQ_ERR_SHAPE;  // last window inside the row
    if (a.R == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a.threads == 1 && !vaeq::dp_wave_supported(a)) return VAEQ_ERR_SHAPE;
    if ((a.threads == 0 || a.threads == 1) && vaeq::dp_wave_supported(a)) return vaeq::launch_dp_wave(a, st);
    switch (a.threads) {
    case 0:
    case 256: return vaeq::launch_dp_lev<256>(a, (size_t)lds, st);
    case 128: return vaeq::launch_dp_lev<128>(a, (size_t)lds, st);
    case 64: return vaeq::launch_dp_lev<64>(a, (size_t)lds, st);
    }
    return VAEQ_ERR_SHAPE;
}

extern "C" int vaeq_dp_step_debug(const vaeq_dp_args *pa, float *gW, float *gh, void *stream)
{
    if (!pa || !gW || !gh) return VAEQ_ERR_NULL;
    if (pa->n_frames != 1 || pa->steps != 1) return VAEQ_ERR_SHAPE;   // one step: the array layouts are those of n_frames = steps = 1
    vaeq_dp_args a = *pa;
    a.dbg_gW = gW;
    a.dbg_gh = gh;
    return vaeq_dp_train(&a, stream);
}

namespace vaeq {
int64_t dp_wave_resident(int B, int M, int n_lev);   // vaeq_dp_wave.hip
template <int NT, int NLEV>
static int64_t resident_generic(size_t lds)
{
    int nb = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return VAEQ_ERR_DEVICE;
    auto k = dp_train_kernel<NT, NLEV>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, NT, lds) != hipSuccess) return VAEQ_ERR_DEVICE;
    return (int64_t)nb * prop.multiProcessorCount;
}
}  // namespace vaeq

extern "C" int64_t vaeq_dp_resident_runs(int32_t B, int32_t sps, int32_t M, int32_t n_lev, int32_t threads)
{
    const int64_t lds = vaeq_dp_lds_bytes(B, sps, M, n_lev);
    if (lds < 0) return lds;
    vaeq_dp_args a = {};
    a.B = B; a.sps = sps; a.M = M; a.n_lev = n_lev; a.stride_sym = B; a.keep_len = B; a.S = 4;
    if ((threads == 0 || threads == 1) && vaeq::dp_wave_supported(a)) return vaeq::dp_wave_resident(B, M, n_lev);
    if (threads == 1) return VAEQ_ERR_SHAPE;
#define VAEQ_RES(NT)                                                          \
    switch (n_lev) {                                                          \
    case 2: return vaeq::resident_generic<NT, 2>((size_t)lds);                \
    case 4: return vaeq::resident_generic<NT, 4>((size_t)lds);                \
    case 8: return vaeq::resident_generic<NT, 8>((size_t)lds);                \
    }                                                                         \
    return VAEQ_ERR_SHAPE;
    switch (threads) {
    case 0:
    case 256: VAEQ_RES(256)
    case 128: VAEQ_RES(128)
    case 64: VAEQ_RES(64)
    }
#undef VAEQ_RES
    return VAEQ_ERR_SHAPE;
}
