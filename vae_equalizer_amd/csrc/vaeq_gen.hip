// vaeq_gen.hip -- seeded on-device dual-polarisation channel simulator: the input producer of the hot path.
//
// Same physical model as optical_DP_channel/shared_funcs.py:65-90 (generate_data_shaping), evaluated for R runs at once:
//   stage 1  vaeq_gen_dp_tx       PCS symbol draw (:76), zero-stuffing (:77), pulse shaping and extra impulse response as ONE
//                                 'valid' FIR with g = h_pulse * h_channel (:56-63, :79-80)            -> sig[R][2][Ls] complex64
//            (FFT over Ls by the caller: torch.fft / hipFFT)
//   stage 2  vaeq_gen_dp_disperse H(f) = R^T diag(e^{j pi tau_pmd f}, e^{-j pi tau_pmd f}) R, times e^{j 2 (pi f)^2 tau_cd}
//                                 (:38-54), per run rotation angle theta[r], in place on the spectrum
//            (inverse FFT by the caller)
//   stage 3  vaeq_gen_dp_finish   sigma_n from the run's mean power (:83), complex AWGN (:84), split into the planar
//                                 rx[R][2][2][sps*N] the training kernel reads (:88)
// Randomness is counter based (Philox4x32-10, key = seed): every value is a pure function of (seed, frame, run, stream, index),
// so frames are reproducible and independent of launch geometry.  The random STREAM differs from numpy's (the reference
// seeds nothing); the numpy restatement in channel.py is the bit-faithful one.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vaeq.h"
#include "vaeq_common.h"

namespace vaeq {

struct Philox4 { uint32_t x, y, z, w; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__host__ __device__ inline float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1), 24 bits

enum { STREAM_SYMBOLS = 0, STREAM_NOISE = 1 };

// level index of symbol n of (run, pol): inverse CDF of the PCS pmf on u ~ U(0,1); x -> I, y -> Q
__device__ __forceinline__ void draw_symbol(uint64_t seed, uint32_t frame, uint32_t run, int pol, uint32_t n, const float *cdf, int n_lev,
                                            int &li, int &lq)
{
    const Philox4 r = philox4x32_10(n, run, frame, (uint32_t)(STREAM_SYMBOLS * 2 + pol), (uint32_t)seed, (uint32_t)(seed >> 32));
    const float ui = u01(r.x), uq = u01(r.y);
    li = 0; lq = 0;
    for (int i = 0; i < n_lev - 1; i++) { li += ui >= cdf[i]; lq += uq >= cdf[i]; }
}

constexpr int TX_TILE = 1024, TX_NT = 256, TX_MAXG = 96;

// stage 1: sig[r][p][s] = sum_k g[k] * up[s + Lg-1-k],  up[j] = symbol[j/sps] if j % sps == 0 else 0   (np.convolve 'valid')
__global__ __launch_bounds__(TX_NT) void gen_tx_kernel(int N_conv, int sps, int n_lev, int Lg, int Ls, const float *__restrict__ amp,
                                                       const float *__restrict__ cdf_g, const float2 *__restrict__ g, uint64_t seed,
                                                       uint32_t frame, int npol, float2 *__restrict__ sig)
{
    __shared__ float2 sym[(TX_TILE + TX_MAXG) / 2 + 4];
    __shared__ float2 gs[TX_MAXG];
    __shared__ float cdf[8];
    const int run = blockIdx.z, pol = blockIdx.y, s0 = blockIdx.x * TX_TILE, tid = threadIdx.x;
    if (tid < n_lev) cdf[tid] = cdf_g[(size_t)run * n_lev + tid];
    for (int i = tid; i < Lg; i += TX_NT) gs[i] = g[i];
    __syncthreads();
    // symbols touched by this tile: up-index j in [s0, s0 + TILE + Lg - 1)  ->  n in [ceil(s0/sps), ...]
    const int nlo = (s0 + sps - 1) / sps, nhi = min(N_conv - 1, (s0 + TX_TILE + Lg - 2) / sps);
    for (int n = nlo + tid; n <= nhi; n += TX_NT) {
        int li, lq;
        draw_symbol(seed, frame, run, pol, n, cdf, n_lev, li, lq);
        sym[n - nlo] = make_float2(amp[li], amp[lq]);
    }
    __syncthreads();
    for (int s = s0 + tid; s < min(Ls, s0 + TX_TILE); s += TX_NT) {
        float ar = 0.f, ai = 0.f;
        // j = s + Lg-1-k must be a multiple of sps: k = (s + Lg - 1) - sps*n
        const int jhi = s + Lg - 1;
        for (int n = (s + sps - 1) / sps; n * sps <= jhi && n < N_conv; n++) {
            const int k = jhi - n * sps;
            const float2 x = sym[n - nlo], c = gs[k];
            ar = fmaf(c.x, x.x, ar); ar = fmaf(-c.y, x.y, ar);
            ai = fmaf(c.x, x.y, ai); ai = fmaf(c.y, x.x, ai);
        }
        sig[((size_t)run * npol + pol) * Ls + s] = make_float2(ar, ai);
    }
}

// TX reference data[r][p][c][n'] = amplitude of symbol n' + lo (shared_funcs.py:89), fp16
__global__ __launch_bounds__(256) void gen_ref_kernel(int N, int lo, int n_lev, const float *__restrict__ amp, const float *__restrict__ cdf_g,
                                                      uint64_t seed, uint32_t frame, int npol, __half *__restrict__ data)
{
    const int run = blockIdx.z, pol = blockIdx.y;
    float cdf[8];
    for (int i = 0; i < n_lev; i++) cdf[i] = cdf_g[(size_t)run * n_lev + i];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        int li, lq;
        draw_symbol(seed, frame, run, pol, n + lo, cdf, n_lev, li, lq);
        data[((size_t)(run * npol + pol) * 2 + 0) * N + n] = __float2half(amp[li]);
        data[((size_t)(run * npol + pol) * 2 + 1) * N + n] = __float2half(amp[lq]);
    }
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// stage 2: spectrum of both polarisations times H(f) and the CD phase (shared_funcs.py:40-53), in place
__global__ __launch_bounds__(256) void gen_disperse_kernel(int Ls, double fs_over_Ls, double tau_cd, double tau_pmd, float2 e0, float2 e1,
                                                           const float *__restrict__ theta, float2 *__restrict__ spec)
{
    const int run = blockIdx.y;
    float st, ct;
    sincosf(theta[run], &st, &ct);
    float2 *X0 = spec + (size_t)run * 2 * Ls, *X1 = X0 + Ls;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < Ls; f += gridDim.x * blockDim.x) {
        const double freq = (double)(f < (Ls + 1) / 2 ? f : f - Ls) * fs_over_Ls;          // np.fft.fftfreq
        const double pf = 3.14159265358979323846 * freq;
        float sc, cc, sd, cd;
        sincosf((float)(2.0 * pf * pf * tau_cd), &sc, &cc);
        sincosf((float)(pf * tau_pmd), &sd, &cd);
        const float2 ecd = make_float2(cc, sc), d = make_float2(cd, sd), di = make_float2(cd, -sd);
        // R = [[c e0, s e0], [-s e1, c e1]],  RT = [[c e0, -s e0], [s e1, c e1]]  (the reference's "R_T", :47-48)
        const float2 ce0 = make_float2(ct * e0.x, ct * e0.y), se0 = make_float2(st * e0.x, st * e0.y);
        const float2 ce1 = make_float2(ct * e1.x, ct * e1.y), se1 = make_float2(st * e1.x, st * e1.y);
        const float2 nse0 = make_float2(-se0.x, -se0.y), nse1 = make_float2(-se1.x, -se1.y);
        const float2 H00 = cadd(cmul(cmul(ce0, d), ce0), cmul(cmul(nse0, di), nse1));
        const float2 H01 = cadd(cmul(cmul(ce0, d), se0), cmul(cmul(nse0, di), ce1));
        const float2 H10 = cadd(cmul(cmul(se1, d), ce0), cmul(cmul(ce1, di), nse1));
        const float2 H11 = cadd(cmul(cmul(se1, d), se0), cmul(cmul(ce1, di), ce1));
        const float2 a = X0[f], b = X1[f];
        X0[f] = cmul(cadd(cmul(H00, a), cmul(H01, b)), ecd);
        X1[f] = cmul(cadd(cmul(H10, a), cmul(H11, b)), ecd);
    }
}

// stage 3a: mean |sig|^2 per run over both polarisations and all Ls samples (:83)
__global__ __launch_bounds__(256) void gen_power_kernel(int Ls, int npol, const float2 *__restrict__ sig, float *__restrict__ power)
{
    __shared__ float red[64];
    const int run = blockIdx.x;
    const float2 *s = sig + (size_t)run * npol * Ls;
    float acc = 0.f;
    for (int i = threadIdx.x; i < npol * Ls; i += 256) acc += s[i].x * s[i].x + s[i].y * s[i].y;
    block_reduce3<256>(acc, 0.f, 0.f, red);
    if (threadIdx.x == 0) power[run] = red[0] / (float)(npol * Ls);
}

// stage 3b: AWGN + planar split: rx[r][p][0/1][s] = Re/Im(sig + sigma_n (n1 + j n2)), s < sps*N   (:84-88)
__global__ __launch_bounds__(256) void gen_finish_kernel(int Ls, int Lout, int sps, const float *__restrict__ snr_db, const float *__restrict__ power,
                                                         uint64_t seed, uint32_t frame, int npol, const float2 *__restrict__ sig,
                                                         float *__restrict__ rx, float *__restrict__ sigma_out)
{
    const int run = blockIdx.z, pol = blockIdx.y;
    const float sigma = sqrtf(power[run] * (float)sps * 0.5f / exp10f(snr_db[run] * 0.1f));
    if (sigma_out && pol == 0 && blockIdx.x == 0 && threadIdx.x == 0) sigma_out[run] = sigma;
    const float2 *s = sig + ((size_t)run * npol + pol) * Ls;
    float *rI = rx + ((size_t)(run * npol + pol) * 2 + 0) * Lout, *rQ = rI + Lout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Lout; i += gridDim.x * blockDim.x) {
        const Philox4 r = philox4x32_10((uint32_t)i, run, frame, (uint32_t)(STREAM_NOISE * 2 + pol), (uint32_t)seed, (uint32_t)(seed >> 32));
        float sn, cs;                                           // Box-Muller: two independent N(0,1)
        const float rad = sqrtf(-2.0f * __logf(u01(r.x)));
        __sincosf(6.283185307179586f * u01(r.y), &sn, &cs);
        rI[i] = s[i].x + sigma * rad * cs;
        rQ[i] = s[i].y + sigma * rad * sn;
    }
}

}  // namespace vaeq

extern "C" int vaeq_gen_dp_tx(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t ref_offset,
                              const float *amp, const float *cdf, const float *g_complex, uint64_t seed, uint32_t frame,
                              float *sig_complex, void *data_f16, void *stream)
{
    if (!amp || !cdf || !g_complex || !sig_complex) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || sps <= 0 || Lg <= 0 || Lg > vaeq::TX_MAXG || !(n_lev == 2 || n_lev == 4 || n_lev == 8) || ref_offset < 0)
        return VAEQ_ERR_SHAPE;
    if (Ls != sps * (N_conv - 1) + 1 - Lg + 1 || ref_offset + N > N_conv || Ls < sps * N) return VAEQ_ERR_SHAPE;   // np.convolve 'valid' length
    if (R == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(vaeq::gen_tx_kernel, dim3((Ls + vaeq::TX_TILE - 1) / vaeq::TX_TILE, 2, R), dim3(vaeq::TX_NT), 0, st, N_conv, sps, n_lev, Lg,
                       Ls, amp, cdf, reinterpret_cast<const float2 *>(g_complex), seed, frame, 2, reinterpret_cast<float2 *>(sig_complex));
    if (data_f16)
        hipLaunchKernelGGL(vaeq::gen_ref_kernel, dim3((N + 255) / 256 > 64 ? 64 : (N + 255) / 256, 2, R), dim3(256), 0, st, N, ref_offset, n_lev,
                           amp, cdf, seed, frame, 2, reinterpret_cast<__half *>(data_f16));
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_gen_dp_disperse(int32_t R, int32_t Ls, double fs, double tau_cd, double tau_pmd, float e0_re, float e0_im, float e1_re,
                                    float e1_im, const float *theta, float *spec_complex, void *stream)
{
    if (!theta || !spec_complex) return VAEQ_ERR_NULL;
    if (R < 0 || Ls <= 0) return VAEQ_ERR_SHAPE;
    if (R == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(vaeq::gen_disperse_kernel, dim3((Ls + 255) / 256, R), dim3(256), 0, st, Ls, fs / (double)Ls, tau_cd, tau_pmd,
                       make_float2(e0_re, e0_im), make_float2(e1_re, e1_im), theta, reinterpret_cast<float2 *>(spec_complex));
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_gen_dp_finish(int32_t R, int32_t N, int32_t sps, int32_t Ls, const float *snr_db, uint64_t seed, uint32_t frame,
                                  const float *sig_complex, float *power_ws, float *rx, float *sigma_out, void *stream)
{
    if (!snr_db || !sig_complex || !power_ws || !rx) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || sps <= 0 || Ls < sps * N) return VAEQ_ERR_SHAPE;
    if (R == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Lout = sps * N;
    hipLaunchKernelGGL(vaeq::gen_power_kernel, dim3(R), dim3(256), 0, st, Ls, 2, reinterpret_cast<const float2 *>(sig_complex), power_ws);
    hipLaunchKernelGGL(vaeq::gen_finish_kernel, dim3((Lout + 255) / 256 > 64 ? 64 : (Lout + 255) / 256, 2, R), dim3(256), 0, st, Ls, Lout, sps,
                       snr_db, power_ws, seed, frame, 2, reinterpret_cast<const float2 *>(sig_complex), rx, sigma_out);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

// Single-polarisation AWGN / ISI channel (AWGN_channel/func_VAELE_MQAM_shaping.py:39-61) for R runs: the same three stages with one
// polarisation and no dispersion step: PCS symbols (:45), zero-stuffing + pulse shaping + channel impulse response as one 'valid'
// FIR with g = rrc * h_channel (:47-52), sigma_n from the mean power (:54), complex AWGN (:55), planar rx[R][2][sps*N] (:57) and the
// TX reference data[R][2][N] (fp16, :59).  sig_ws: [R][Ls] complex64 scratch, power_ws: [R] floats.
extern "C" int vaeq_gen_awgn(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t ref_offset,
                             const float *amp, const float *cdf, const float *g_complex, const float *snr_db, uint64_t seed, uint32_t frame,
                             float *sig_ws, float *power_ws, float *rx, void *data_f16, float *sigma_out, void *stream)
{
    if (!amp || !cdf || !g_complex || !snr_db || !sig_ws || !power_ws || !rx) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || sps <= 0 || Lg <= 0 || Lg > vaeq::TX_MAXG || !(n_lev == 2 || n_lev == 4 || n_lev == 8) || ref_offset < 0)
        return VAEQ_ERR_SHAPE;
    if (Ls != sps * (N_conv - 1) + 1 - Lg + 1 || ref_offset + N > N_conv || Ls < sps * N) return VAEQ_ERR_SHAPE;
    if (R == 0) return VAEQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Lout = sps * N;
    float2 *sig = reinterpret_cast<float2 *>(sig_ws);
    hipLaunchKernelGGL(vaeq::gen_tx_kernel, dim3((Ls + vaeq::TX_TILE - 1) / vaeq::TX_TILE, 1, R), dim3(vaeq::TX_NT), 0, st, N_conv, sps, n_lev, Lg,
                       Ls, amp, cdf, reinterpret_cast<const float2 *>(g_complex), seed, frame, 1, sig);
    if (data_f16)
        hipLaunchKernelGGL(vaeq::gen_ref_kernel, dim3((N + 255) / 256 > 64 ? 64 : (N + 255) / 256, 1, R), dim3(256), 0, st, N, ref_offset, n_lev,
                           amp, cdf, seed, frame, 1, reinterpret_cast<__half *>(data_f16));
    hipLaunchKernelGGL(vaeq::gen_power_kernel, dim3(R), dim3(256), 0, st, Ls, 1, sig, power_ws);
    hipLaunchKernelGGL(vaeq::gen_finish_kernel, dim3((Lout + 255) / 256 > 64 ? 64 : (Lout + 255) / 256, 1, R), dim3(256), 0, st, Ls, Lout, sps,
                       snr_db, power_ws, seed, frame, 1, sig, rx, sigma_out);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}
