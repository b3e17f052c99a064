// vaeq_wave.h -- building blocks shared by the wave-per-run kernels (vaeq_dp_wave.hip, vaeq_awgn_wave.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace vaeq {

// One wave per workgroup: LDS instructions of a wave execute in issue order, so a later ds_read of any lane sees an earlier
// ds_write of any other lane without a hardware wait.  What is needed is only that the COMPILER keeps that order --
// __syncthreads() would also emit s_waitcnt vmcnt(0) and stall every phase on the step's in-flight q/y stores.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("" ::: "memory"); }

// NW wavefronts per run: phases are separated by s_barrier after an LDS-only wait -- the in-flight global stores of a step still
// never stall a phase (gfx950 backs off barriers; the compiler adds no vmcnt(0) in front of this one).
template <int NW>
__device__ __forceinline__ void sync_lds()
{
    if constexpr (NW == 1) wave_lds_sync();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sum K per-wave values (already reduced over the wave's lanes) over the NW waves of a run, same order in every wave; red: K NW floats.
template <int NW, int K>
__device__ __forceinline__ void waves_sum(float (&v)[K], float *red, int lane, int wv)
{
    if constexpr (NW > 1) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < K; k++) red[k * NW + wv] = v[k];
        }
        sync_lds<NW>();
#pragma unroll
        for (int k = 0; k < K; k++) {
            float t = red[k * NW];
#pragma unroll
            for (int w = 1; w < NW; w++) t += red[k * NW + w];
            v[k] = t;
        }
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));   // element-wise ops compile to v_pk_{add,mul,fma}_f32

// Complex MACs.  A complex accumulator is kept as TWO packed partial sums,  a = sum re(t) * v  and  b = sum im(t) * v  (t = tap,
// v = sample), so that every MAC is exactly two v_pk_fma_f32 whose scalar factor rides on op_sel: no swap, no negation, no
// v_mov in the inner loops.  The two are combined once at the end:  t * v = (a.x - b.y, a.y + b.x),  v * conj(t) = (a.x + b.y, a.y - b.x).
struct cacc { v2f a, b; };
__device__ __forceinline__ cacc cacc0() { return cacc{v2f{0.f, 0.f}, v2f{0.f, 0.f}}; }
__device__ __forceinline__ void cmac(cacc &c, float tr, float ti, float2 v)           // c += (tr, ti) (x) v
{
    const v2f vv = {v.x, v.y};
    c.a += tr * vv;
    c.b += ti * vv;
}
__device__ __forceinline__ float2 cfin(const cacc &c) { return make_float2(c.a.x - c.b.y, c.a.y + c.b.x); }    // sum t * v
__device__ __forceinline__ float2 cfinc(const cacc &c) { return make_float2(c.a.x + c.b.y, c.a.y - c.b.x); }   // sum v * conj(t)

__device__ __forceinline__ float wave_incl_scan(float v, int lane)                    // inclusive prefix sum over lanes
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// polyphase-4 row length (in float2) for a sample-rate array of `len` cells: phase arrays end up 16 (mod 64) dwords apart, so
// the stride-4 reads of consecutive lanes are bank-conflict free
__host__ __device__ inline int wave_lph(int len)
{
    int lph = (len + 3) / 4 + 1;
    while ((lph & 31) != 8 && (lph & 31) != 24) lph++;
    return lph;
}

}  // namespace vaeq
