// vaeq_misc.hip -- stand-alone soft demapper, inference-mode butterfly FIR, version / error strings.
#include <hip/hip_runtime.h>
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

}  // namespace vaeq

extern "C" int vaeq_soft_demap(int32_t R, int64_t N, int32_t n_lev, const float *y, const float *amp, const float *var,
                               const float *nu_sc, float *q, void *stream)
{
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
