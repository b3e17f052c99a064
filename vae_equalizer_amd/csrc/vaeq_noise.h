// vaeq_noise.h -- the counter-based random numbers of the on-device channel simulators (vaeq_gen.hip) and the AWGN noise built from them,
// shared with the kernels that add the noise where the samples are consumed (vaeq_awgn_validate_gen, vaeq_awgn.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Complex AWGN of the sample pair (2j, 2j + 1) of (run, pol) in one frame: ONE Philox call, two Box-Muller transforms
// (func_VAELE_MQAM_shaping.py:55: sigma_n * (randn + 1j * randn)); v0 / v1 = the clean samples, noisy on return.  The additions are explicit
// fused multiply-adds so that every kernel that applies this noise produces the same bits.
__device__ __forceinline__ void awgn_noise_pair(uint32_t j, uint32_t run, uint32_t frame, int pol, uint64_t seed, float sigma, float2 &v0, float2 &v1)
{
    const Philox4 r = philox4x32_10(j, run, frame, (uint32_t)(STREAM_NOISE * 2 + pol), (uint32_t)seed, (uint32_t)(seed >> 32));
    float sn0, cs0, sn1, cs1;
    const float rad0 = sigma * sqrtf(-2.0f * __logf(u01(r.x))), rad1 = sigma * sqrtf(-2.0f * __logf(u01(r.z)));
    __sincosf(6.283185307179586f * u01(r.y), &sn0, &cs0);
    __sincosf(6.283185307179586f * u01(r.w), &sn1, &cs1);
    v0.x = __fmaf_rn(rad0, cs0, v0.x);
    v0.y = __fmaf_rn(rad0, sn0, v0.y);
    v1.x = __fmaf_rn(rad1, cs1, v1.x);
    v1.y = __fmaf_rn(rad1, sn1, v1.y);
}

// |v|^2 onto a running power sum, as explicit fused multiply-adds: the one-pass, two-pass and clean-frame forms of the AWGN generator must agree
// on sigma_n to the bit, whatever else their loops do around this expression
__device__ __forceinline__ float awgn_power_add(float pw, float2 v) { return __fmaf_rn(v.y, v.y, __fmaf_rn(v.x, v.x, pw)); }

// sigma_n of one run from the power sums of its 2048-sample tiles (added in tile order: every kernel gets the same bits) and the run's SNR
// (:54: sqrt(mean |x|^2 * sps / 2 / 10^(SNR/10)))
__device__ __forceinline__ float awgn_sigma_from_parts(const float *part, int n_parts, int Ls, int sps, float snr_db)
{
    float pw = 0.f;
    for (int t = 0; t < n_parts; t++) pw += part[t];
    return sqrtf(pw / (float)Ls * (float)sps * 0.5f / exp10f(snr_db * 0.1f));
}

}  // namespace vaeq
