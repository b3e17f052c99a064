// vaeq_misc.hip -- stand-alone soft demapper, inference-mode butterfly FIR, version / error strings.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <stdint.h>

#include "vaeq.h"
#include "vaeq_common.h"

namespace vaeq {

// soft_dec (shared_funcs.py:529-542): one thread per (run, pol, I/Q, symbol); q rows written coalesced along N.
template <int NLEV>
__global__ __launch_bounds__(256) void soft_demap_kernel(int64_t N, const float *__restrict__ y, const float *__restrict__ amp_g,
                                                         const float *__restrict__ var, const float *__restrict__ nu_sc,
                                                         float *__restrict__ q)
{
    const int run = blockIdx.z, oc = blockIdx.y, o = oc >> 1, c = oc & 1;
    float amp[NLEV], amp2[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; amp2[i] = amp[i] * amp[i]; }
    const float i2v = 0.5f / var[run * 2 + o], nusc = nu_sc[run];
    const float *yr = y + ((size_t)run * 4 + oc) * N;
    float *qr = q + ((size_t)run * 4 * NLEV + (size_t)o * 2 * NLEV + c * NLEV) * N;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float qq[NLEV];
        soft_demap<NLEV>(yr[n], amp, amp2, i2v, nusc, qq);
#pragma unroll
        for (int i = 0; i < NLEV; i++) qr[(size_t)i * N + n] = qq[i];
    }
}

// twoXtwoFIR.forward without training (shared_funcs.py:500-527): one thread per (run, o, symbol).
// Taps of the run in LDS; samples straight from global (L1/L2 absorb the 2*M-fold reuse).
template <int NLEV>
__global__ __launch_bounds__(256) void dp_forward_kernel(int64_t N, int sps, int M, const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ amp_g, const float *__restrict__ var,
                                                         const float *__restrict__ nu_sc, float *__restrict__ q, float *__restrict__ yout)
{
    __shared__ float Ws[8 * 64];
    const int run = blockIdx.z, o = blockIdx.y;
    for (int i = threadIdx.x; i < 8 * M; i += blockDim.x) Ws[i] = W[(size_t)run * 8 * M + i];
    __syncthreads();
    float amp[NLEV], amp2[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; amp2[i] = amp[i] * amp[i]; }
    const float i2v = 0.5f / var[run * 2 + o], nusc = nu_sc[run];
    const int64_t L = N * sps;
    const int mh = M / 2;
    const float *xb = x + (size_t)run * 4 * L;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float yI = 0.f, yQ = 0.f;
        for (int p = 0; p < 2; p++) {
            const float *xr = xb + (size_t)(p * 2 + 0) * L, *xi = xb + (size_t)(p * 2 + 1) * L;
            const float *wr = Ws + (o * 4 + p) * M, *wi = Ws + (o * 4 + 2 + p) * M;
            for (int k = 0; k < M; k++) {
                const int64_t s = n * sps + k - mh;
                if (s < 0 || s >= L) continue;
                const float a_ = xr[s], b_ = xi[s];
                yI = fmaf(wr[k], a_, yI);
                yI = fmaf(-wi[k], b_, yI);
                yQ = fmaf(wr[k], b_, yQ);
                yQ = fmaf(wi[k], a_, yQ);
            }
        }
        yout[((size_t)run * 4 + o * 2 + 0) * N + n] = yI;
        yout[((size_t)run * 4 + o * 2 + 1) * N + n] = yQ;
        if (q) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                float qq[NLEV];
                soft_demap<NLEV>(c ? yQ : yI, amp, amp2, i2v, nusc, qq);
#pragma unroll
                for (int i = 0; i < NLEV; i++) q[((size_t)run * 4 * NLEV + (size_t)o * 2 * NLEV + c * NLEV + i) * N + n] = qq[i];
            }
        }
    }
}

// loss_function_shaping alone (shared_funcs.py:92-137): one workgroup per run, moments of q in LDS.
// Values only -- the training path never calls this (its forward is fused into dp_train_kernel).
template <int NLEV>
__global__ __launch_bounds__(256) void dp_loss_kernel(int B, int sps, int M, const float *__restrict__ q, const float *__restrict__ x,
                                                      const float *__restrict__ h, const float *__restrict__ amp_g,
                                                      const float *__restrict__ P, float *__restrict__ loss, float *__restrict__ var_est)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    const int run = blockIdx.x, tid = threadIdx.x;
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh;
    float *mu = sm, *vr = sm + 4 * B, *hs = vr + 4 * B, *VS = hs + 8 * M, *red = VS + 2 * M;
    float amp[NLEV], invP[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; invP[i] = 1.0f / P[(size_t)run * NLEV + i]; }
    const float *qr = q + (size_t)run * 4 * NLEV * B, *xr = x + (size_t)run * 4 * L;
    for (int i = tid; i < 8 * M; i += 256) hs[i] = h[(size_t)run * 8 * M + i];
    float kl = 0.f;
    for (int it = tid; it < 4 * B; it += 256) {
        const int vc = it / B, n = it - vc * B;
        float qq[NLEV], m1 = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) { qq[i] = qr[((size_t)vc * NLEV + i) * B + n]; m1 = fmaf(amp[i], qq[i], m1); }
        float m2 = 0.f;
        const bool inr = n >= mh && n < B - mh;
#pragma unroll
        for (int i = 0; i < NLEV; i++) {
            const float d = amp[i] - m1;
            m2 = fmaf(qq[i] * d, d, m2);
            if (inr) kl = fmaf(qq[i], __logf(qq[i] * invP[i] + 1e-12f), kl);
        }
        mu[it] = m1;
        vr[it] = m2;
    }
    __syncthreads();
    float se0 = 0.f, se1 = 0.f;
    for (int it = tid; it < 2 * nm; it += 256) {
        const int chi = it / nm, t = it - chi * nm;
        float dr = 0.f, di = 0.f;
        for (int v = 0; v < 2; v++) {
            const float *hr = hs + ((chi * 2 + v) * 2 + 0) * M, *hi = hr + M;
            for (int j = (t + Mh) % sps; j <= Mh; j += sps) {
                const int np = (t + Mh - j) / sps;
                const float a_ = mu[(v * 2 + 0) * B + np], b_ = mu[(v * 2 + 1) * B + np];
                dr = fmaf(hr[j], a_, dr); dr = fmaf(-hi[j], b_, dr);
                di = fmaf(hi[j], a_, di); di = fmaf(hr[j], b_, di);
            }
        }
        const float er = xr[(size_t)(chi * 2 + 0) * L + mh + t] - dr, ei = xr[(size_t)(chi * 2 + 1) * L + mh + t] - di;
        if (chi) se1 += er * er + ei * ei; else se0 += er * er + ei * ei;
    }
    for (int it = tid; it < 2 * M; it += 256) {
        const int v = it / M, j = it - v * M;
        const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
        float acc = 0.f;
        for (int np = lo; np <= hi_; np++) acc += vr[(v * 2 + 0) * B + np] + vr[(v * 2 + 1) * B + np];
        VS[it] = acc;
    }
    block_reduce3<256>(se0, se1, kl, red);
    if (tid == 0) {
        float C0 = red[0], C1 = red[1];
        for (int i = 0; i < 2 * M; i++) {
            const int v = i / M, j = i - v * M;
            const float a0 = hs[((0 * 2 + v) * 2 + 0) * M + j], b0 = hs[((0 * 2 + v) * 2 + 1) * M + j];
            const float a1 = hs[((1 * 2 + v) * 2 + 0) * M + j], b1 = hs[((1 * 2 + v) * 2 + 1) * M + j];
            C0 = fmaf(a0 * a0 + b0 * b0, VS[i], C0);
            C1 = fmaf(a1 * a1 + b1 * b1, VS[i], C1);
        }
        loss[run] = (float)nm * (logf(C0) + logf(C1)) + red[2];
        var_est[run * 2 + 0] = C0 / (float)nm;
        var_est[run * 2 + 1] = C1 / (float)nm;
    }
}

}  // namespace vaeq


extern "C" int vaeq_soft_demap(int32_t R, int64_t N, int32_t n_lev, const float *y, const float *amp, const float *var,
                               const float *nu_sc, float *q, void *stream)
{
    if (R == 0 || N == 0) return VAEQ_OK;                      // an empty batch owns no memory: its pointers may be NULL
    if (!y || !amp || !var || !nu_sc || !q) return VAEQ_ERR_NULL;
    if (R < 0 || N < 0) return VAEQ_ERR_SHAPE;
    if (R == 0 || N == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((N + 255) / 256 > 4096 ? 4096 : (N + 255) / 256), 4, R);
    switch (n_lev) {
    case 2: hipLaunchKernelGGL(vaeq::soft_demap_kernel<2>, grid, dim3(256), 0, st, N, y, amp, var, nu_sc, q); break;
    case 4: hipLaunchKernelGGL(vaeq::soft_demap_kernel<4>, grid, dim3(256), 0, st, N, y, amp, var, nu_sc, q); break;
    case 8: hipLaunchKernelGGL(vaeq::soft_demap_kernel<8>, grid, dim3(256), 0, st, N, y, amp, var, nu_sc, q); break;
    default: return VAEQ_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_dp_forward(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *W,
                               const float *amp, const float *var, const float *nu_sc, float *q, float *y, void *stream)
{
    if (R == 0 || N == 0) return VAEQ_OK;                      // an empty batch owns no memory: its pointers may be NULL
    if (!x || !W || !amp || !var || !nu_sc || !y) return VAEQ_ERR_NULL;
    if (R < 0 || N < 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63) return VAEQ_ERR_SHAPE;
    if (R == 0 || N == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((N + 255) / 256 > 4096 ? 4096 : (N + 255) / 256), 2, R);
    switch (n_lev) {
    case 2: hipLaunchKernelGGL(vaeq::dp_forward_kernel<2>, grid, dim3(256), 0, st, N, sps, M, x, W, amp, var, nu_sc, q, y); break;
    case 4: hipLaunchKernelGGL(vaeq::dp_forward_kernel<4>, grid, dim3(256), 0, st, N, sps, M, x, W, amp, var, nu_sc, q, y); break;
    case 8: hipLaunchKernelGGL(vaeq::dp_forward_kernel<8>, grid, dim3(256), 0, st, N, sps, M, x, W, amp, var, nu_sc, q, y); break;
    default: return VAEQ_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_dp_loss(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x,
                            const float *h, const float *amp, const float *P, float *loss, float *var_est, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!q || !x || !h || !amp || !P || !loss || !var_est) return VAEQ_ERR_NULL;
    if (R < 0 || B <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || B * sps - 2 * (M / 2) <= 0 || B <= 2 * (M / 2)) return VAEQ_ERR_SHAPE;
    const size_t lds = sizeof(float) * (size_t)(8 * B + 8 * M + 2 * M + 64);
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VAEQ_LOSS_CASE(NL)                                                                                                         \
    case NL: {                                                                                                                     \
        auto k = vaeq::dp_loss_kernel<NL>;                                                                                         \
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess)                                                        \
            return VAEQ_ERR_LDS;                                                                                                   \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, B, sps, M, q, x, h, amp, P, loss, var_est);                             \
    } break;
    switch (n_lev) {
        VAEQ_LOSS_CASE(2)
        VAEQ_LOSS_CASE(4)
        VAEQ_LOSS_CASE(8)
    default: return VAEQ_ERR_SHAPE;
    }
#undef VAEQ_LOSS_CASE
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

namespace vaeq {
static thread_local char g_last_kernel[160] = "";
void note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof g_last_kernel, fmt, ap);
    va_end(ap);
}

// grid-stride 16-byte copy: the measured HBM copy bandwidth next to the spec peak
__global__ __launch_bounds__(256) void stream_copy_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace vaeq

extern "C" int vaeq_last_kernel(char *buf, int32_t len)
{
    if (!buf || len <= 0) return VAEQ_ERR_NULL;
    snprintf(buf, (size_t)len, "%s", vaeq::g_last_kernel);
    return VAEQ_OK;
}

extern "C" int vaeq_stream_copy(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (!dst || !src) return VAEQ_ERR_NULL;
    if (bytes < 0 || (bytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) || (reinterpret_cast<uintptr_t>(src) & 15)) return VAEQ_ERR_SHAPE;
    if (bytes == 0) return VAEQ_OK;
    hipLaunchKernelGGL(vaeq::stream_copy_kernel, dim3(256 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<float4 *>(dst), reinterpret_cast<const float4 *>(src), (size_t)bytes / 16);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_version(void) { return VAEQ_VERSION; }

extern "C" const char *vaeq_strerror(int code)
{
    switch (code) {
    case VAEQ_OK: return "ok";
    case VAEQ_ERR_NULL: return "required pointer is NULL";
    case VAEQ_ERR_SHAPE: return "inconsistent or unsupported sizes";
    case VAEQ_ERR_LDS: return "per-run working set exceeds the 160 KiB LDS of a CU";
    case VAEQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case VAEQ_ERR_DEVICE: return "no gfx950 device";
    }
    return "unknown vaeq error";
}

// ======================================================================================================================
// Backward passes of the two stand-alone operators, so that a reference-style loop
//     q, out = net(x, ...); loss, _ = loss_function_shaping(q, x, h_est, ...); loss.backward(); optimizer.step()
// runs on HIP kernels through torch.autograd.Function wrappers (vae_equalizer_amd/autograd_ops.py).  One workgroup per run.
// Not the training hot path (that is the fused vaeq_dp_train) -- the differentiable form of the drop-in operator surface.
namespace vaeq {

// d loss / d q and d loss / d h_est of loss_function_shaping (shared_funcs.py:92-137), times the upstream gradient g_up[run].
template <int NLEV>
__global__ __launch_bounds__(256) void dp_loss_bwd_kernel(int B, int sps, int M, const float *__restrict__ q, const float *__restrict__ x,
                                                          const float *__restrict__ h, const float *__restrict__ amp_g,
                                                          const float *__restrict__ P, const float *__restrict__ g_up,
                                                          float *__restrict__ gq, float *__restrict__ gh)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    const int run = blockIdx.x, tid = threadIdx.x;
    const int L = B * sps, mh = M / 2, Mh = 2 * mh, nm = L - Mh;
    float *mu = sm, *vr = mu + 4 * B, *es = vr + 4 * B, *hs = es + 4 * nm, *VS = hs + 8 * M, *red = VS + 2 * M;
    float amp[NLEV], invP[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) { amp[i] = amp_g[i]; invP[i] = 1.0f / P[(size_t)run * NLEV + i]; }
    const float *qr = q + (size_t)run * 4 * NLEV * B, *xr = x + (size_t)run * 4 * L;
    float *gqr = gq + (size_t)run * 4 * NLEV * B;
    const float up = g_up[run];
    for (int i = tid; i < 8 * M; i += 256) hs[i] = h[(size_t)run * 8 * M + i];
    for (int it = tid; it < 4 * B; it += 256) {                 // moments of q (:107-113)
        const int vc = it / B, n = it - vc * B;
        float qq[NLEV], m1 = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) { qq[i] = qr[((size_t)vc * NLEV + i) * B + n]; m1 = fmaf(amp[i], qq[i], m1); }
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) { const float d = amp[i] - m1; m2 = fmaf(qq[i] * d, d, m2); }
        mu[it] = m1;
        vr[it] = m2;
    }
    __syncthreads();
    float se0 = 0.f, se1 = 0.f;
    for (int it = tid; it < 2 * nm; it += 256) {                // e = x - D (:123-127)
        const int chi = it / nm, t = it - chi * nm;
        float dr = 0.f, di = 0.f;
        for (int v = 0; v < 2; v++) {
            const float *hr = hs + ((chi * 2 + v) * 2 + 0) * M, *hi = hr + M;
            for (int j = (t + Mh) % sps; j <= Mh; j += sps) {
                const int np = (t + Mh - j) / sps;
                const float a_ = mu[(v * 2 + 0) * B + np], b_ = mu[(v * 2 + 1) * B + np];
                dr = fmaf(hr[j], a_, dr); dr = fmaf(-hi[j], b_, dr);
                di = fmaf(hi[j], a_, di); di = fmaf(hr[j], b_, di);
            }
        }
        const float er = xr[(size_t)(chi * 2 + 0) * L + mh + t] - dr, ei = xr[(size_t)(chi * 2 + 1) * L + mh + t] - di;
        es[(chi * 2 + 0) * nm + t] = er;
        es[(chi * 2 + 1) * nm + t] = ei;
        if (chi) se1 += er * er + ei * ei; else se0 += er * er + ei * ei;
    }
    for (int it = tid; it < 2 * M; it += 256) {                 // VS (:128)
        const int v = it / M, j = it - v * M;
        const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
        float acc = 0.f;
        for (int np = lo; np <= hi_; np++) acc += vr[(v * 2 + 0) * B + np] + vr[(v * 2 + 1) * B + np];
        VS[it] = acc;
    }
    block_reduce3<256>(se0, se1, 0.f, red);
    float C0 = red[0], C1 = red[1];
    for (int i = 0; i < 2 * M; i++) {
        const int v = i / M, j = i - v * M;
        const float a0 = hs[((0 * 2 + v) * 2 + 0) * M + j], b0 = hs[((0 * 2 + v) * 2 + 1) * M + j];
        const float a1 = hs[((1 * 2 + v) * 2 + 0) * M + j], b1 = hs[((1 * 2 + v) * 2 + 1) * M + j];
        C0 = fmaf(a0 * a0 + b0 * b0, VS[i], C0);
        C1 = fmaf(a1 * a1 + b1 * b1, VS[i], C1);
    }
    const float gC0 = up * (float)nm / C0, gC1 = up * (float)nm / C1;
    for (int it = tid; it < 4 * M; it += 256) {                 // d/dh
        const int cv = it / M, j = it - cv * M, chi = cv >> 1, v = cv & 1;
        const int lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
        const float *er = es + (chi * 2 + 0) * nm, *ei = er + nm;
        float ar = 0.f, ai = 0.f;
        for (int np = lo; np <= hi_; np++) {
            const int t = np * sps - Mh + j;
            const float c_ = mu[(v * 2 + 0) * B + np], d_ = mu[(v * 2 + 1) * B + np];
            ar = fmaf(er[t], c_, ar); ar = fmaf(ei[t], d_, ar);
            ai = fmaf(ei[t], c_, ai); ai = fmaf(-er[t], d_, ai);
        }
        const float gC = chi ? gC1 : gC0, vs = VS[v * M + j];
        const int ir = (cv * 2 + 0) * M + j, ii = ir + M;
        gh[(size_t)run * 8 * M + ir] = gC * (-2.0f * ar + 2.0f * hs[ir] * vs);
        gh[(size_t)run * 8 * M + ii] = gC * (-2.0f * ai + 2.0f * hs[ii] * vs);
    }
    for (int it = tid; it < 2 * B; it += 256) {                 // d/dq through mu, rho and the KL term
        const int v = it / B, n = it - v * B, sx = n * sps;
        const int jlo = max(0, Mh - sx), jhi = min(Mh, nm - 1 + Mh - sx);
        float ur = 0.f, ui = 0.f, gv = 0.f;
        for (int chi = 0; chi < 2; chi++) {
            const float *er = es + (chi * 2 + 0) * nm + (sx - Mh), *ei = er + nm;
            const float *hr = hs + ((chi * 2 + v) * 2 + 0) * M, *hi = hr + M;
            float pr = 0.f, pi = 0.f, ph = 0.f;
            for (int j = jlo; j <= jhi; j++) {
                pr = fmaf(er[j], hr[j], pr); pr = fmaf(ei[j], hi[j], pr);
                pi = fmaf(ei[j], hr[j], pi); pi = fmaf(-er[j], hi[j], pi);
                ph = fmaf(hr[j], hr[j], ph); ph = fmaf(hi[j], hi[j], ph);
            }
            const float gC = chi ? gC1 : gC0;
            ur = fmaf(-2.0f * gC, pr, ur);
            ui = fmaf(-2.0f * gC, pi, ui);
            gv = fmaf(gC, ph, gv);
        }
        const bool inr = n >= mh && n < B - mh;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int ix = (v * 2 + c) * B + n;
            const float A = (c ? ui : ur) - 2.0f * mu[ix] * gv;  // dL/dmu (Var = rho - mu^2, :113)
#pragma unroll
            for (int i = 0; i < NLEV; i++) {
                const size_t qi = ((size_t)(v * 2 + c) * NLEV + i) * B + n;
                float g = amp[i] * A + amp[i] * amp[i] * gv;
                if (inr) {
                    const float r = qr[qi] * invP[i], re = r + 1e-12f;
                    g += up * (logf(re) + r / re);               // d/dq [q log(q/P + eps)]  (:131-132)
                }
                gqr[qi] = g;
            }
        }
    }
}

// d loss / d W of twoXtwoFIR.forward (shared_funcs.py:500-527) given d loss / d q and (optionally) d loss / d out.
template <int NLEV>
__global__ __launch_bounds__(256) void dp_forward_bwd_kernel(int N, int sps, int M, const float *__restrict__ x, const float *__restrict__ q,
                                                             const float *__restrict__ y, const float *__restrict__ gq, const float *__restrict__ gy_in,
                                                             const float *__restrict__ amp_g, const float *__restrict__ var,
                                                             float *__restrict__ gW)
{
    extern __shared__ float4 smem4[];
    float *gy = reinterpret_cast<float *>(smem4);               // [2][2][N]
    const int run = blockIdx.x, tid = threadIdx.x, L = N * sps, mh = M / 2;
    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = amp_g[i];
    const float *qr = q + (size_t)run * 4 * NLEV * N, *gqr = gq + (size_t)run * 4 * NLEV * N, *yr = y + (size_t)run * 4 * N;
    const float *xr = x + (size_t)run * 4 * L;
    for (int it = tid; it < 4 * N; it += 256) {                 // softmin backward (:521-523): dz_i = q_i (gq_i - sum q gq), dz_i/dy = -(y-a_i)/var
        const int oc = it / N, n = it - oc * N, o = oc >> 1;
        float qq[NLEV], dot = 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) { qq[i] = qr[((size_t)oc * NLEV + i) * N + n]; dot = fmaf(qq[i], gqr[((size_t)oc * NLEV + i) * N + n], dot); }
        const float yy = yr[(size_t)oc * N + n], iv = 1.0f / var[run * 2 + o];
        float g = gy_in ? gy_in[((size_t)run * 4 + oc) * N + n] : 0.f;
#pragma unroll
        for (int i = 0; i < NLEV; i++) g = fmaf(qq[i] * (gqr[((size_t)oc * NLEV + i) * N + n] - dot), -(yy - amp[i]) * iv, g);
        gy[it] = g;
    }
    __syncthreads();
    for (int it = tid; it < 4 * M; it += 256) {                 // conv weight gradient through the channel packing (:505,507)
        const int op = it / M, k = it - op * M, o = op >> 1, p = op & 1;
        float ar = 0.f, ai = 0.f;
        for (int n = 0; n < N; n++) {
            const int s = n * sps + k - mh;
            if (s < 0 || s >= L) continue;
            const float a_ = gy[(o * 2 + 0) * N + n], b_ = gy[(o * 2 + 1) * N + n], c_ = xr[(size_t)(p * 2 + 0) * L + s], d_ = xr[(size_t)(p * 2 + 1) * L + s];
            ar = fmaf(a_, c_, ar); ar = fmaf(b_, d_, ar);
            ai = fmaf(b_, c_, ai); ai = fmaf(-a_, d_, ai);
        }
        gW[(size_t)run * 8 * M + (o * 4 + p) * M + k] = ar;
        gW[(size_t)run * 8 * M + (o * 4 + 2 + p) * M + k] = ai;
    }
}

}  // namespace vaeq

extern "C" int vaeq_dp_loss_bwd(int32_t R, int32_t B, int32_t sps, int32_t M, int32_t n_lev, const float *q, const float *x, const float *h,
                                const float *amp, const float *P, const float *g_up, float *gq, float *gh, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!q || !x || !h || !amp || !P || !g_up || !gq || !gh) return VAEQ_ERR_NULL;
    if (R < 0 || B <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || B * sps - 2 * (M / 2) <= 0 || B <= 2 * (M / 2)) return VAEQ_ERR_SHAPE;
    const size_t lds = sizeof(float) * (size_t)(8 * B + 4 * (B * sps - 2 * (M / 2)) + 10 * M + 64);
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VAEQ_LB_CASE(NL)                                                                                                           \
    case NL: {                                                                                                                     \
        auto k = vaeq::dp_loss_bwd_kernel<NL>;                                                                                     \
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess)                                                        \
            return VAEQ_ERR_LDS;                                                                                                   \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, B, sps, M, q, x, h, amp, P, g_up, gq, gh);                             \
    } break;
    switch (n_lev) {
        VAEQ_LB_CASE(2)
        VAEQ_LB_CASE(4)
        VAEQ_LB_CASE(8)
    default: return VAEQ_ERR_SHAPE;
    }
#undef VAEQ_LB_CASE
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_dp_forward_bwd(int32_t R, int32_t N, int32_t sps, int32_t M, int32_t n_lev, const float *x, const float *q, const float *y,
                                   const float *gq, const float *gy, const float *amp, const float *var, float *gW, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!x || !q || !y || !gq || !amp || !var || !gW) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63) return VAEQ_ERR_SHAPE;
    const size_t lds = sizeof(float) * (size_t)4 * N;
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VAEQ_FB_CASE(NL)                                                                                                           \
    case NL: {                                                                                                                     \
        auto k = vaeq::dp_forward_bwd_kernel<NL>;                                                                                  \
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess)                                                        \
            return VAEQ_ERR_LDS;                                                                                                   \
        hipLaunchKernelGGL(k, dim3(R), dim3(256), lds, st, N, sps, M, x, q, y, gq, gy, amp, var, gW);                             \
    } break;
    switch (n_lev) {
        VAEQ_FB_CASE(2)
        VAEQ_FB_CASE(4)
        VAEQ_FB_CASE(8)
    default: return VAEQ_ERR_SHAPE;
    }
#undef VAEQ_FB_CASE
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}
