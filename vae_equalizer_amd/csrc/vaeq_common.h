// vaeq_common.h -- device helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace vaeq {

// Host side: every training-kernel launcher records the instantiation it launched (vaeq_last_kernel reports it, so a benchmark names
// the kernel it actually timed instead of a literal that can go stale when the dispatch changes).  Defined in vaeq_misc.hip.
void note_kernel(const char *fmt, ...);

// Four floats at any 4-byte boundary: one global_load_dwordx4 / global_store_dwordx4 (gfx950 serves unaligned vector accesses to global memory).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// Sum over the 64 lanes of a wave, result in every lane.  Fixed xor-butterfly order:
// bitwise reproducible run to run (no float atomics anywhere in this library).
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Workgroup sums of three values; results land in red[0..2] for every thread.
// red needs 3*(NT/64)+3 floats.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void block_reduce3(float a, float b, float c, float *red)
{
    constexpr int NW = NT / 64;
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if (NW == 1) {
        if (threadIdx.x == 0) { red[0] = a; red[1] = b; red[2] = c; }
        __syncthreads();
        return;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[4 + w * 3 + 0] = a; red[4 + w * 3 + 1] = b; red[4 + w * 3 + 2] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < NW; i++) { s0 += red[4 + i * 3]; s1 += red[4 + i * 3 + 1]; s2 += red[4 + i * 3 + 2]; }
        red[0] = s0; red[1] = s1; red[2] = s2;
    }
    __syncthreads();
}

// torch.optim.Adam (single-tensor path, amsgrad=False) for one parameter:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); p.addcdiv_(m, sqrt(v)/sqrt(bc2)+eps, -lr/bc1)
// step_size = lr/bc1 and bc2s = sqrt(bc2) are prepared in double by the caller.
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, float step_size, float bc2s)
{
    m = fmaf(g - m, 0.1f, m);
    v = v * 0.999f;
    v = v + (0.001f * g) * g;
    const float denom = sqrtf(v) / bc2s + 1e-8f;
    p = p + (-step_size * m) / denom;
}

// Same update with the hardware reciprocal / square root (1 ulp each) instead of the IEEE-exact expansions: the step
// changes by ~2e-7 relative, i.e. < 1e-9 absolute on a tap -- far below the fp32 noise floor of the gradients.
__device__ __forceinline__ void adam_update_fast(float &p, float &m, float &v, float g, float step_size, float rbc2s)
{
    m = fmaf(g - m, 0.1f, m);
    v = v * 0.999f;
    v = v + (0.001f * g) * g;
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), rbc2s, 1e-8f);
    p = fmaf(-step_size * m, __builtin_amdgcn_rcpf(denom), p);
}

// ... and the AMSGrad form with the hardware reciprocal / square root (the VAE-NN kernel's 1650 parameters per step; rbc2s = 1 / sqrt(1 - beta2^t))
__device__ __forceinline__ void adam_update_amsgrad_fast(float &p, float &m, float &v, float &vmax, float g, float step_size, float rbc2s)
{
    m = fmaf(g - m, 0.1f, m);
    v = v * 0.999f;
    v = v + (0.001f * g) * g;
    vmax = fmaxf(vmax, v);
    const float denom = fmaf(__builtin_amdgcn_sqrtf(vmax), rbc2s, 1e-8f);
    p = fmaf(-step_size * m, __builtin_amdgcn_rcpf(denom), p);
}

__device__ __forceinline__ void adam_update_amsgrad(float &p, float &m, float &v, float &vmax, float g, float step_size, float bc2s)
{
    m = fmaf(g - m, 0.1f, m);
    v = v * 0.999f;
    v = v + (0.001f * g) * g;
    vmax = fmaxf(vmax, v);
    const float denom = sqrtf(vmax) / bc2s + 1e-8f;
    p = p + (-step_size * m) / denom;
}

// per-axis soft demapper (shared_funcs.py:521-523 / 529-542): q_i = softmax_i(-(y-a_i)^2/(2 var) - nu_sc a_i^2)
template <int NLEV>
__device__ __forceinline__ void soft_demap(float y, const float (&amp)[NLEV], const float (&amp2)[NLEV], float i2v, float nusc, float (&q)[NLEV])
{
    float zmax = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NLEV; i++) {
        const float d = y - amp[i];
        q[i] = -(d * d * i2v + nusc * amp2[i]);
        zmax = fmaxf(zmax, q[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NLEV; i++) {
        q[i] = __expf(q[i] - zmax);
        s += q[i];
    }
    const float rs = 1.0f / s;
#pragma unroll
    for (int i = 0; i < NLEV; i++) q[i] *= rs;
}

}  // namespace vaeq
