// vaeq_wave.h -- building blocks shared by the wave-per-run kernels (vaeq_dp_wave.hip, vaeq_awgn_wave.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#ifndef VAEQ_SUM_BCAST
#define VAEQ_SUM_BCAST 1                               // wave sums finish on two DPP broadcasts + one v_readlane (0: four v_readlane + three adds; A/B knob)
#endif

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
// The same MAC with tap and sample as 8-byte register pairs straight from LDS (lds2): the tap's two halves ride on op_sel of the pair, so
// nothing is moved.  cmul starts an accumulator (first tap of a sum) instead of adding to a zeroed one: no v_mov 0 per accumulator.
__device__ __forceinline__ void cmac(cacc &c, v2f t, v2f v)
{
    c.a += t.x * v;
    c.b += t.y * v;
}
__device__ __forceinline__ void cmul(cacc &c, v2f t, v2f v)
{
    c.a = t.x * v;
    c.b = t.y * v;
}
template <bool FIRST>
__device__ __forceinline__ void cmacf(cacc &c, v2f t, v2f v)
{
    if constexpr (FIRST) cmul(c, t, v);
    else cmac(c, t, v);
}
// One 8-byte LDS read that stays one ds_read_b64 (volatile: never fused with a neighbour).  gfx950 serves ds_read_b64 at 256 B/clk but the
// fused ds_read2_b64 at 128 B/clk (8 instead of 2 x 2 LDS cycles per wave, MI355X_MICROARCH.md LDS table), its 8-bit offsets cost extra address
// adds, and the backend copies the 4th dword of a 16-byte result before it can broadcast it into a v_pk_fma_f32.
// (the cast names the LDS address space: address-space inference leaves volatile accesses alone, they would become flat loads)
typedef const volatile __attribute__((address_space(3))) v2f lds_cv2f;
__device__ __forceinline__ v2f lds2(const float2 *p) { return *(lds_cv2f *)p; }
__device__ __forceinline__ v2f lds2(const float *p) { return *(lds_cv2f *)p; }
__device__ __forceinline__ float2 f2(v2f v) { return make_float2(v.x, v.y); }
// lds2 where the kernel can afford it (V), an ordinary load (free to fuse / reorder: fewer live registers) where it is register-starved
template <bool V>
__device__ __forceinline__ v2f lds2_if(const float2 *p)
{
    if constexpr (V) return lds2(p);
    const float2 t = *p;
    return v2f{t.x, t.y};
}

// Buffer addressing for the streamed rows: address = base (4 SGPRs, one descriptor per array and frame) + a per-lane byte offset (VGPR) + a
// per-row byte offset (SGPR, folded into the instruction): NO vector ALU work per store / load, no 64-bit row pointers to keep (or spill) in
// scalar registers.  A lane that must not store / load passes OOB as its offset: the access is dropped by the range check (loads return 0).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);   // raw buffer, gfx9 dword 3
}
__device__ __forceinline__ void bst64(v2f v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bst32(float v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bst16(unsigned short v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b16(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bst8(unsigned char v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b8(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float4 bld128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(float4, v);
}

// Two-deep software pipeline over n >= 1 stages: load(i, r) issues stage i's NR independent 8-byte LDS reads, fma(i, r, first) consumes them.
// Stage i + 1 is in flight while stage i computes (register sets A / B alternate, no copies), so a wave hides its own LDS latency instead of
// relying on the one other wave of its SIMD to be in an FMA burst at that moment.  Loads run up to two stages past the end (never consumed):
// the caller's load(i) must stay inside the LDS allocation for i < n + 2 -- that keeps the loop body free of conditional loads.
// PIPE = false: plain loop (load, compute, load, compute ...) for phases that cannot spare the second register set.
template <int NR, bool INIT, bool PIPE = true, class L, class F>
__device__ __forceinline__ void pipe2(int n, L load, F fma)
{
    if (!INIT && n <= 0) return;                       // (INIT: the caller guarantees n >= 1, stage 0 starts the accumulators)
    if constexpr (!PIPE) {
        v2f A[NR];
        load(0, A);
        fma(0, A, std::integral_constant<bool, INIT>{});
#pragma unroll 1
        for (int i = 1; i < n; i++) {
            load(i, A);
            fma(i, A, std::false_type{});
        }
        return;
    }
    v2f A[NR], B[NR];
    load(0, A);
    load(1, B);
    fma(0, A, std::integral_constant<bool, INIT>{});
    int i = 1;
#pragma unroll 1
    for (; i + 1 < n; i += 2) {                        // B holds stage i
        load(i + 1, A);
        fma(i, B, std::false_type{});
        load(i + 2, B);
        fma(i + 1, A, std::false_type{});
    }
    if (i < n) fma(i, B, std::false_type{});
}

// Final combine of the two packed partial sums, ONE v_pk_add_f32 each: the swap of b's halves and the sign ride on op_sel / neg (the backend
// does not find this form: it shuffles the halves with v_mov and adds them as scalars).
__device__ __forceinline__ v2f cfin2(const cacc &c)                                    // sum t * v = (a.x - b.y, a.y + b.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(c.a), "v"(c.b));
    return r;
}
__device__ __forceinline__ v2f cfinc2(const cacc &c)                                   // sum v * conj(t) = (a.x + b.y, a.y - b.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(c.a), "v"(c.b));
    return r;
}
__device__ __forceinline__ float2 cfin(const cacc &c) { const v2f r = cfin2(c); return make_float2(r.x, r.y); }
__device__ __forceinline__ float2 cfinc(const cacc &c) { const v2f r = cfinc2(c); return make_float2(r.x, r.y); }

// Cross-lane sums on the vector ALU (DPP), no LDS round trips: a ds_bpermute butterfly costs six dependent LDS latencies (~100 cycles each under
// load) during which the wave issues nothing; the DP kernel has five such sums and three scans per step.  Fixed order => bitwise reproducible.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v)           // lanes the masks switch off (or whose source lies outside the row) contribute 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v)    // sum over the 64 lanes, result in every lane (uniform)
{
    v += dpp_f<0xB1>(v);                                   // quad_perm:[1,0,3,2]
    v += dpp_f<0x4E>(v);                                   // quad_perm:[2,3,0,1]
    v += dpp_f<0x141>(v);                                  // row_half_mirror
    v += dpp_f<0x140>(v);                                  // row_mirror: every lane of a 16-lane row holds the row's sum
#if VAEQ_SUM_BCAST
    // the four row sums r0..r3 meet in lane 63 as (r3 + r2) + (r1 + r0) -- the same additions as below (IEEE addition commutes: bitwise the same
    // result) with two DPP broadcasts and ONE v_readlane instead of four v_readlane and three adds
    v += dpp_f<0x142, 0xA>(v);                             // row_bcast:15 into rows 1 and 3: r1 + r0, r3 + r2
    v += dpp_f<0x143, 0xC>(v);                             // row_bcast:31 into rows 2 and 3: lane 63 = (r3 + r2) + (r1 + r0)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float row_incl_scan_dpp(float v)   // inclusive prefix sum inside each 16-lane row
{
    float s = v + dpp_f<0x111>(v);                         // row_shr:1
    s += dpp_f<0x112>(v);                                  // row_shr:2
    s += dpp_f<0x113>(v);                                  // row_shr:3   -> sums of (up to) 4
    s += dpp_f<0x114, 0xF, 0xE>(s);                        // row_shr:4 into banks 1..3
    s += dpp_f<0x118, 0xF, 0xC>(s);                        // row_shr:8 into banks 2, 3
    return s;
}
__device__ __forceinline__ float half_incl_scan_dpp(float v)  // inclusive prefix sum inside each 32-lane half
{
    float s = row_incl_scan_dpp(v);
    s += dpp_f<0x142, 0xA>(s);                             // row_bcast:15 into rows 1 and 3
    return s;
}
__device__ __forceinline__ float wave_incl_scan_dpp(float v)  // inclusive prefix sum over the 64 lanes
{
    float s = half_incl_scan_dpp(v);
    s += dpp_f<0x143, 0xC>(s);                             // row_bcast:31 into rows 2 and 3
    return s;
}

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
