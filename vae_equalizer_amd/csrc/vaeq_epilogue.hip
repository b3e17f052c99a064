// vaeq_epilogue.hip -- per-frame epilogue of the DP runs on the device, one workgroup per run.
//
// Restates, for R runs at once,
//   shared_funcs.py:290-338  find_shift / find_shift_symb_full  (|correlation| of TX I and Q of both polarisations with the
//                            equaliser-side sequence rolled by -10..10 symbols; straight vs swapped pairing)
//   shared_funcs.py:188-222  SER_IQflip          (argmax(q) against TX, min over IQ-flip x 4 quadrant rotations)
//   shared_funcs.py:225-287  SER_constell_shaping + dec_on_bound (PCS-aware thresholds on the mean-radius-normalised FIR output)
// and the roll / cut / slice logic of func_VAELE_DP_MQAM_shaping.py:68-89 (batch_len > 0: the last shift[0]+10 symbols of every
// minibatch and 11 (+max|shift|) symbols at the frame ends are dropped) resp. func_VAEflex_DP_MQAM_shaping.py:72-84 (batch_len = 0).
// Data-dependent rolls and slices are index arithmetic and masks; counters are integers (LDS atomics), float sums are reduced in
// a fixed order, so results are bitwise reproducible.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "vaeq.h"
#include "vaeq_common.h"
#include "vaeq_wave.h"

namespace vaeq {

#ifndef EPI_WAVES
#define EPI_WAVES 6
#endif
#ifndef EPI_WAVES_CMA
#define EPI_WAVES_CMA 6                       // (5 waves per SIMD without the CMA mode's 92 bytes of scratch: 6.8 vs 6.6 ms, no gain)
#endif
constexpr int EPI_NT = 256, N_SHIFT = 21, HALF_SHIFT = 10, N_CUT = 10, EDGE = 11;
constexpr int N_CHUNK = EPI_NT / 8;           // thread = (chunk, E-polarisation b, TX polarisation a, lag half h): 32 x 2 x 2 x 2; the I and Q rows of a ride in one packed FMA
constexpr int EPI_CH = 22;                    // symbols per chunk (even: a thread's 32-sample lag window is 16 aligned register pairs)
constexpr int N_LAGH = 11;                    // lags per thread: h = 0 -> lags 0..10, h = 1 -> lags 10..20 (lag 10 is computed by both, taken from h = 0)
constexpr int EPI_TILE = N_CHUNK * EPI_CH;    // 704 symbols staged in LDS per correlation tile
constexpr int ES_LEN = EPI_TILE + 2 * HALF_SHIFT + 4;

struct EpiShared {
    float corr[2][2][2][N_SHIFT];             // [c][b][a][lag]
    int shift[2];
    int r;
    int cnt[16];                              // error counts [hypothesis 0..7][pol]
    int kept;
    float lo[8], hi[8];                       // decision interval of each TX level (shared_funcs.py:234-236)
    float red[64];
    union {
        struct {
            float2 txs[2 * EPI_TILE];                 // TX tile: [symbol][a] = (I, Q) of polarisation a
            alignas(16) float es[2][ES_LEN];          // equaliser-side tile with a 10-symbol halo on both sides
        } t;
        float2 part[N_CHUNK][2][2][N_SHIFT];          // per-chunk partial correlations (after the last tile)
    } u;
};

// correlation of the TX reference with E[b][.] rolled by lag-10 (shared_funcs.py:300-304).  The symbol axis is walked in tiles staged
// in LDS; a thread owns one (chunk of 22 symbols, b, TX polarisation a, half of the lags) and both TX rows (I, Q) of a: its lag window
// (32 samples of E) sits in 16 register pairs, and per symbol one 8-byte TX read feeds 11 packed FMAs whose E factor is one half of a
// pair (picked by op_sel at compile time: no moves).
__device__ __forceinline__ void epi_correlate(const float *__restrict__ E /*[2][N] or strided*/, int64_t estride, const __half *__restrict__ tx, int64_t N64,
                              EpiShared &sh)
{
    const int tid = threadIdx.x, N = (int)N64;
    const int chunk = tid >> 3, b = (tid >> 2) & 1, a = (tid >> 1) & 1, h = tid & 1;
    const bool wide_tx = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(tx) & 7) == 0;
    v2f acc[N_LAGH];
#pragma unroll
    for (int i = 0; i < N_LAGH; i++) acc[i] = v2f{0.f, 0.f};
#ifdef EPI_SKIP_CORR
    for (int t0 = 0; t0 < EPI_TILE; t0 += EPI_TILE) {
#else
    for (int t0 = 0; t0 < N; t0 += EPI_TILE) {
#endif
        // staging in groups of four consecutive symbols: 8- and 16-byte loads wherever the group lies inside the row, no integer division
        for (int g = tid; g < EPI_TILE / 4; g += EPI_NT) {     // TX rows; symbols past the end contribute zeros
            const int n = t0 + 4 * g;
            float v[4][4];
            if (wide_tx && n + 3 < N) {
#pragma unroll
                for (int row = 0; row < 4; row++) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(tx + (size_t)row * N + n);
                    const __half2 h01 = *reinterpret_cast<const __half2 *>(&w.x), h23 = *reinterpret_cast<const __half2 *>(&w.y);
                    v[row][0] = __low2float(h01); v[row][1] = __high2float(h01); v[row][2] = __low2float(h23); v[row][3] = __high2float(h23);
                }
            } else {
#pragma unroll
                for (int row = 0; row < 4; row++)
#pragma unroll
                    for (int e = 0; e < 4; e++) v[row][e] = n + e < N ? __half2float(tx[(size_t)row * N + n + e]) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sh.u.t.txs[2 * (4 * g + e) + 0] = make_float2(v[0][e], v[1][e]);
                sh.u.t.txs[2 * (4 * g + e) + 1] = make_float2(v[2][e], v[3][e]);
            }
        }
        constexpr int EG = (EPI_TILE + 2 * HALF_SHIFT) / 4;    // E[b][t0 - 10 .. t0 + TILE + 10), indices mod N
        for (int i = tid; i < 2 * EG; i += EPI_NT) {
            const int bb = i >= EG, j = 4 * (i - bb * EG), m0 = t0 + j - HALF_SHIFT;
            const float *row = E + (int64_t)bb * estride;
            float4 v;
            if (m0 >= 0 && m0 + 3 < N) {
                const f4u w = *reinterpret_cast<const f4u *>(row + m0);
                v = make_float4(w.x, w.y, w.z, w.w);
            } else {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int m = m0 + e;
                    while (m < 0) m += N;
                    while (m >= N) m -= N;
                    t[e] = row[m];
                }
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
            *reinterpret_cast<float4 *>(&sh.u.t.es[bb][j]) = v;
        }
        __syncthreads();
        {
            // roll(E, lag-10)[n] = E[n - (lag-10)] -> es index j + 20 - lag: symbol j0 + s and lag l meet sample es[j0 + s + 20 - l].  With the
            // thread's lags l = lb + i (lb = 10 h) and its window W[k] = es[j0 + 10 (1 - h) + k] that is W[s + 10 - i] for both halves.
            const float *eb = sh.u.t.es[b] + chunk * EPI_CH + HALF_SHIFT * (1 - h);
            const float2 *tp = sh.u.t.txs + 2 * (chunk * EPI_CH) + a;
            v2f W[16];
#pragma unroll
            for (int k = 0; k < 16; k++) W[k] = lds2(eb + 2 * k);
#pragma unroll
            for (int s = 0; s < EPI_CH; s++) {
                const v2f tv = lds2(tp + 2 * s);
#pragma unroll
                for (int i = 0; i < N_LAGH; i++) {
                    const int e = s + HALF_SHIFT - i;
                    acc[i] += tv * ((e & 1) ? W[e >> 1].y : W[e >> 1].x);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < N_LAGH; i++)
        if (h == 0 || i > 0) sh.u.part[chunk][b][a][HALF_SHIFT * h + i] = make_float2(acc[i].x, acc[i].y);
    __syncthreads();
    if (tid < 2 * 2 * N_SHIFT) {                                // fixed-order sum over chunks
        const int l = tid % N_SHIFT, ba = tid / N_SHIFT, b2 = ba >> 1, a2 = ba & 1;
        v2f tot = {0.f, 0.f};
        for (int k = 0; k < N_CHUNK; k++) tot += lds2(&sh.u.part[k][b2][a2][l]);
        sh.corr[0][b2][a2][l] = fabsf(tot.x);
        sh.corr[1][b2][a2][l] = fabsf(tot.y);
    }
    __syncthreads();
    if (tid == 0) {                                             // shared_funcs.py:303-314
        float cm[2][2];
        int pick[2][2];
        for (int b = 0; b < 2; b++)
            for (int a = 0; a < 2; a++) {
                float best[2];
                int bi[2];
                for (int c = 0; c < 2; c++) {
                    best[c] = sh.corr[c][b][a][0];
                    bi[c] = 0;
                    for (int l = 1; l < N_SHIFT; l++)
                        if (sh.corr[c][b][a][l] > best[c]) { best[c] = sh.corr[c][b][a][l]; bi[c] = l; }
                }
                const int cw = best[1] > best[0] ? 1 : 0;       // torch.max over (I, Q): first maximum wins ties
                cm[b][a] = best[cw];
                pick[b][a] = bi[cw];
            }
        const bool straight = (cm[0][0] + cm[1][1]) >= (cm[0][1] + cm[1][0]);
        sh.shift[0] = HALF_SHIFT - (straight ? pick[0][0] : pick[0][1]);
        sh.shift[1] = HALF_SHIFT - (straight ? pick[1][1] : pick[1][0]);
        sh.r = straight ? 0 : 1;
        for (int i = 0; i < 16; i++) sh.cnt[i] = 0;
        sh.kept = 0;
    }
    __syncthreads();
}

// Add a thread's count to a workgroup counter in LDS: wave sum on the vector ALU first (exact: counts < 2^24), then ONE integer LDS atomic per wave.
// A plain atomicAdd per thread is turned by the compiler's atomic optimiser into a serial loop over the wave's 64 lanes (s_ff1 / v_readlane / add per
// lane): 17 counters x 64 lanes x ~8 instructions per walk and wavefront -- that loop, not the walk's arithmetic or its memory traffic, was what the
// SER walks of this kernel spent most of their time in (phase stamps: 150 k of 180 k cycles; profiles/r03).  Integer sums: order-independent, bitwise reproducible.
__device__ __forceinline__ void epi_count_add(int *dst, int v)
{
    const int tot = (int)wave_sum_dpp((float)v);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, tot);
}

// symbols that survive the per-minibatch cut (:73-77) and the frame-edge slice (:79)
__device__ __forceinline__ bool epi_keep(int n, int N, int batch_len, int shift0, int ms)
{
    if (batch_len <= 0) return n >= EDGE && n < N - EDGE - ms;
    int Lk = batch_len - shift0 - N_CUT;
    Lk = Lk < 0 ? 0 : (Lk > batch_len ? batch_len : Lk);
    const int mb = n / batch_len, j = n - mb * batch_len, k = mb * Lk + j, K = (N / batch_len) * Lk;
    return j < Lk && k >= EDGE && k < K - EDGE - ms;
}

// epi_keep for the symbols n = tid, tid + 256, ... of one thread without an integer division per symbol: (minibatch, offset) advance by
// (256 / batch_len, 256 % batch_len) with one carry
struct KeepWalk {
    int N, batch_len, ms, Lk, K, mb, j, dq, dr;
    __device__ __forceinline__ KeepWalk(int n0, int N_, int batch_len_, int shift0, int ms_) : N(N_), batch_len(batch_len_), ms(ms_)
    {
        Lk = K = mb = j = dq = dr = 0;
        if (batch_len > 0) {
            Lk = batch_len - shift0 - N_CUT;
            Lk = Lk < 0 ? 0 : (Lk > batch_len ? batch_len : Lk);
            K = (N / batch_len) * Lk;
            mb = n0 / batch_len; j = n0 - mb * batch_len;
            dq = EPI_NT / batch_len; dr = EPI_NT - dq * batch_len;
        }
    }
    __device__ __forceinline__ bool keep(int n) const
    {
        if (batch_len <= 0) return n >= EDGE && n < N - EDGE - ms;
        const int k = mb * Lk + j;
        return j < Lk && k >= EDGE && k < K - EDGE - ms;
    }
    __device__ __forceinline__ void next()
    {
        mb += dq; j += dr;
        if (j >= batch_len) { j -= batch_len; mb++; }
    }
};

// the same for a thread that owns groups of four consecutive symbols n0 = 4 (tid + 256 i): the group's (minibatch, offset) advance by 1024 symbols
// per step; the members' follow with at most a few carries
struct KeepWalk4 {
    int N, batch_len, ms, Lk, K, mb, j, dq, dr;
    __device__ __forceinline__ KeepWalk4(int n0, int N_, int batch_len_, int shift0, int ms_) : N(N_), batch_len(batch_len_), ms(ms_)
    {
        Lk = K = mb = j = dq = dr = 0;
        if (batch_len > 0) {
            Lk = batch_len - shift0 - N_CUT;
            Lk = Lk < 0 ? 0 : (Lk > batch_len ? batch_len : Lk);
            K = (N / batch_len) * Lk;
            mb = n0 / batch_len; j = n0 - mb * batch_len;
            dq = (4 * EPI_NT) / batch_len; dr = 4 * EPI_NT - dq * batch_len;
        }
    }
    __device__ __forceinline__ void keep4(int n0, bool (&kp)[4]) const
    {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int n = n0 + e;
            if (batch_len <= 0) kp[e] = n >= EDGE && n < N - EDGE - ms;
            else {
                int je = j + e, me = mb;
                while (je >= batch_len) { je -= batch_len; me++; }
                const int k = me * Lk + je;
                kp[e] = n < N && je < Lk && k >= EDGE && k < K - EDGE - ms;
            }
        }
    }
    __device__ __forceinline__ void next()
    {
        mb += dq; j += dr;
        if (j >= batch_len) { j -= batch_len; mb++; }
    }
};


// an LDS read the compiler leaves where the source puts it (a plain read that feeds a short-circuit test is sunk into an exec-masked branch of its own
// behind its own s_waitcnt lgkmcnt(0): tools/scan_isa.py counted 283 such waits for 426 reads in this kernel)
typedef const volatile __attribute__((address_space(3))) float epi_lds_cvf;
__device__ __forceinline__ float epi_ldsv(const float *p) { return *(epi_lds_cvf *)p; }

// four consecutive elements of a row: one wide load when the group lies inside the row (and the row is aligned for it), else the kept members one by one
// (a kept symbol's partner index n + shift never leaves the row: 11 <= n, |shift| <= 10, and the window ends 11 + max|shift| before the row does)
__device__ __forceinline__ void epi_ld_tx(const __half *row, int n0, bool wide, const bool (&kp)[4], float (&out)[4])
{
    if (wide) {
        const uint2 w = *reinterpret_cast<const uint2 *>(row + n0);
        const __half2 h01 = *reinterpret_cast<const __half2 *>(&w.x), h23 = *reinterpret_cast<const __half2 *>(&w.y);
        out[0] = __low2float(h01); out[1] = __high2float(h01); out[2] = __low2float(h23); out[3] = __high2float(h23);
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = kp[e] ? __half2float(row[n0 + e]) : 0.f;
    }
}
__device__ __forceinline__ void epi_ld_y(const float *row, int m0, bool wide, const bool (&kp)[4], float (&out)[4])
{
    if (wide) {
        const f4u v = *reinterpret_cast<const f4u *>(row + m0);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = kp[e] ? row[m0 + e] : 0.f;
    }
}
__device__ __forceinline__ void epi_ld_d(const int8_t *row, int m0, bool wide, const bool (&kp)[4], float (&out)[4])
{
    if (wide) {
        uint32_t w;
        __builtin_memcpy(&w, row + m0, 4);
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = (float)(int8_t)(w >> (8 * e));
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = kp[e] ? (float)row[m0 + e] : 0.f;
    }
}

// CMA = true: the two-stage epilogue of the constant-modulus baselines (func_CMA_DP_MQAM_shaping.py:41-52 and its CMAbatch / CMAflex twins) on the
// phase-corrected output y: the constellation path runs FIRST; its alignment and mean-radius normalisation are then written into y's kept window (the
// reference normalises a view in place, shared_funcs.py:242), the soft demapper (:48) turns that into E_q[x_I] and hard decisions in the workspace, and
// the soft-demapper path runs on those -- its shifts relative to the already aligned sequence, as in the reference.
template <int NLEV, bool CMA = false>
__global__ __launch_bounds__(EPI_NT, CMA ? EPI_WAVES_CMA : EPI_WAVES) void dp_epilogue_kernel(int64_t N, int batch_len, const float *__restrict__ q, const float *__restrict__ y,
                                                             const __half *__restrict__ txg, const float *__restrict__ amp_g,
                                                             const float *__restrict__ var, const float *__restrict__ nu_sc,
                                                             float *__restrict__ ser, int32_t *__restrict__ shift_out,
                                                             int32_t *__restrict__ r_out, float *__restrict__ wsE, int8_t *__restrict__ wsD)
{
    __shared__ EpiShared sh;
    const int run = blockIdx.x, tid = threadIdx.x;
    const float *qr = q ? q + (size_t)run * 4 * NLEV * N : nullptr, *yr = y + (size_t)run * 4 * N;
    const __half *tx = txg + (size_t)run * 4 * N;              // [a][c][n]
    float *E = wsE + (size_t)run * 2 * N;
    int8_t *D = wsD + (size_t)run * 4 * N;
    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = amp_g[i];
    const float scale = 0.5f * (NLEV - 1);
    if (tid == 0) {
        const float thr_scale = 1.0f + 2.0f * nu_sc[run] * var[run * 2 + 0];      // :234
        for (int i = 0; i < NLEV; i++) {
            sh.lo[i] = i == 0 ? -INFINITY : thr_scale * (amp_g[i - 1] + amp_g[i]) / 2;
            sh.hi[i] = i == NLEV - 1 ? INFINITY : thr_scale * (amp_g[i] + amp_g[i + 1]) / 2;
        }
    }

    // ---- pass 1: E_q[x_I] per polarisation (shared_funcs.py:296-297) and hard decisions argmax(q) per axis (:201)
    //      (q == nullptr: the training kernel already wrote both -- vaeq_dp_args.eq_out / dec_out)
    for (int64_t n = tid; q && n < N; n += EPI_NT) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                float best = qr[(size_t)(p * 2 * NLEV + c * NLEV) * N + n];
                int bi = 0;
                if (c == 0) e = amp[0] * best;
#pragma unroll
                for (int i = 1; i < NLEV; i++) {
                    const float v = qr[(size_t)(p * 2 * NLEV + c * NLEV + i) * N + n];
                    if (c == 0) e = fmaf(amp[i], v, e);
                    if (v > best) { best = v; bi = i; }
                }
                D[(size_t)(p * 2 + c) * N + n] = (int8_t)bi;
            }
            E[(size_t)p * N + n] = e;
        }
    }
    __syncthreads();                                            // the workgroup's own global writes are visible to it after the barrier

    for (int pi = 0; pi < 2; pi++) {                            // 0: soft-demapper path on q, 1: constellation path on y
        const int path = CMA ? 1 - pi : pi;
        if (path == 0) epi_correlate(E, N, tx, N, sh);
        else epi_correlate(yr, 2 * N, tx, N, sh);               // y[:, 0, :]: rows 0 and 2 of y[2][2][N]   (:321)
        const int s0 = sh.shift[0], s1 = sh.shift[1], r = sh.r;
        const int ms = max(abs(s0), abs(s1));
        if (tid == 0) {
            shift_out[(size_t)run * 4 + path * 2 + 0] = s0;
            shift_out[(size_t)run * 4 + path * 2 + 1] = s1;
            r_out[(size_t)run * 2 + path] = r;
        }
        // both passes below walk the symbols in groups of four consecutive ones per thread: 8- and 16-byte loads instead of 2- and 4-byte ones
        const bool row_wide = ((int)N & 3) == 0 && (reinterpret_cast<uintptr_t>(tx) & 7) == 0;
        const int NG = ((int)N + 3) >> 2;
        float fac = 1.0f;
        if (path == 1) {                                        // mean radius of TX over mean radius of the aligned output (:242)
            float st = 0.f, sy = 0.f;
            KeepWalk4 kw(4 * tid, (int)N, batch_len, s0, ms);
            for (int g = tid; g < NG; g += EPI_NT, kw.next()) {
                const int n0 = 4 * g;
                bool kp[4];
                kw.keep4(n0, kp);
                if (!(kp[0] || kp[1] || kp[2] || kp[3])) continue;
                // the access form (one wide load per row / the kept members one by one) is a compile-time choice of the group's body: as a run-time flag inside
                // the loaders every load sat in a branch of its own behind its own s_waitcnt vmcnt(0) (182 of the kernel's 186 global loads)
                const bool wide_g = row_wide && n0 >= 12 && n0 + 14 <= (int)N;
                auto body = [&](auto wtag) {
                constexpr bool wide = decltype(wtag)::value;
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int sp = (p - r) & 1, m0 = n0 + (p ? s1 : s0);
                    float ti[4], tq[4], yi[4], yq[4];
                    epi_ld_tx(tx + (size_t)(p * 2 + 0) * N, n0, wide, kp, ti);
                    epi_ld_tx(tx + (size_t)(p * 2 + 1) * N, n0, wide, kp, tq);
                    epi_ld_y(yr + (size_t)(sp * 2 + 0) * N, m0, wide, kp, yi);
                    epi_ld_y(yr + (size_t)(sp * 2 + 1) * N, m0, wide, kp, yq);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (kp[e]) {
                            st += sqrtf(fmaf(ti[e], ti[e], tq[e] * tq[e]));   // (explicit fma: the same rounding as dp_epilogue_lds_kernel's table)
                            sy += sqrtf(fmaf(yi[e], yi[e], yq[e] * yq[e]));
                        }
                }
                };
                if (wide_g) body(std::true_type{}); else body(std::false_type{});
            }
            block_reduce3<EPI_NT>(st, sy, 0.f, sh.red);
            fac = sh.red[0] / sh.red[1];
            __syncthreads();
        }
        int cnt[16];
#pragma unroll
        for (int i = 0; i < 16; i++) cnt[i] = 0;
        int kept = 0;
        KeepWalk4 kw(4 * tid, (int)N, batch_len, s0, ms);
#ifdef EPI_SKIP_SER
        for (int g = tid; g < NG / 64; g += EPI_NT, kw.next()) {
#else
        for (int g = tid; g < NG; g += EPI_NT, kw.next()) {
#endif
            const int n0 = 4 * g;
            bool kp[4];
            kw.keep4(n0, kp);
            if (!(kp[0] || kp[1] || kp[2] || kp[3])) continue;
            kept += (int)kp[0] + (int)kp[1] + (int)kp[2] + (int)kp[3];
            const bool wide_g = row_wide && n0 >= 12 && n0 + 14 <= (int)N;
            auto body = [&](auto wtag) {                         // (access form as a compile-time choice: see the radius walk above)
            constexpr bool wide = decltype(wtag)::value;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int sp = (p - r) & 1;                     // roll(r, 0): row p comes from row p - r  (:71)
                const int m0 = n0 + (p ? s1 : s0);              // roll(-shift): out[n] = in[n + shift]     (:72)
                float ti[4], tq[4], u0[4], u1[4];
                epi_ld_tx(tx + (size_t)(p * 2 + 0) * N, n0, wide, kp, ti);
                epi_ld_tx(tx + (size_t)(p * 2 + 1) * N, n0, wide, kp, tq);
                if (path == 0) {
                    epi_ld_d(D + (size_t)(sp * 2 + 0) * N, m0, wide, kp, u0);
                    epi_ld_d(D + (size_t)(sp * 2 + 1) * N, m0, wide, kp, u1);
                } else {
                    epi_ld_y(yr + (size_t)(sp * 2 + 0) * N, m0, wide, kp, u0);
                    epi_ld_y(yr + (size_t)(sp * 2 + 1) * N, m0, wide, kp, u1);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (!kp[e]) continue;
                    const float dI = rintf(scale * ti[e] + scale);       // :198
                    const float dQ = rintf(scale * tq[e] + scale);
                    const float dQi = -(dQ - 2.0f * scale);         // IQ flip (:199)
                    if (path == 0) {
                        const float a0 = u0[e], a1 = u1[e];
                        // decisions under rotation by 0, pi, pi/2, 3pi/2 (:201-217)
                        const float hI[4] = {a0, -(a0 - 2.0f * scale), -(a1 - 2.0f * scale), a1};
                        const float hQ[4] = {a1, -(a1 - 2.0f * scale), a0, -(a0 - 2.0f * scale)};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            cnt[(2 * k + 0) * 2 + p] += (dI != hI[k]) || (dQ != hQ[k]);
                            cnt[(2 * k + 1) * 2 + p] += (dI != hI[k]) || (dQi != hQ[k]);
                        }
                    } else {
                        const float yi = u0[e] * fac, yq = u1[e] * fac;
                        const float rI[4] = {yi, -yi, -yq, yq}, rQ[4] = {yq, -yq, yi, -yi};        // :245-262
                        // d_vec0[lev] <= v < d_vec1[lev] (:267-287): the bounds of the symbol's three TX levels are read ONCE, up front
                        const int lI = min(max((int)dI, 0), NLEV - 1), lQ = min(max((int)dQ, 0), NLEV - 1), lQi = min(max((int)dQi, 0), NLEV - 1);
                        const float loI = epi_ldsv(sh.lo + lI), hiI = epi_ldsv(sh.hi + lI), loQ = epi_ldsv(sh.lo + lQ), hiQ = epi_ldsv(sh.hi + lQ);
                        const float loQi = epi_ldsv(sh.lo + lQi), hiQi = epi_ldsv(sh.hi + lQi);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool okI = (loI <= rI[k]) & (rI[k] < hiI);
                            const bool okQ = (loQ <= rQ[k]) & (rQ[k] < hiQ), okQi = (loQi <= rQ[k]) & (rQ[k] < hiQi);
                            cnt[(2 * k + 0) * 2 + p] += !(okI & okQ);
                            cnt[(2 * k + 1) * 2 + p] += !(okI & okQi);
                        }
                    }
                }
            }
            };
            if (wide_g) body(std::true_type{}); else body(std::false_type{});
        }
#pragma unroll
        for (int i = 0; i < 16; i++) epi_count_add(&sh.cnt[i], cnt[i]);
        epi_count_add(&sh.kept, kept);
        __syncthreads();
        if (tid < 2) {                                          // min over the 8 hypotheses (:221 / :264)
            const float den = (float)max(sh.kept, 1);
            float best = 2.0f;
            for (int k = 0; k < 8; k++) best = fminf(best, (float)sh.cnt[k * 2 + tid] / den);
            ser[(size_t)run * 4 + (path == 0 ? 2 : 0) + tid] = best;       // rows 0-1 constellation, 2-3 soft demapper (:79,89)
        }
        if (CMA && path == 1) {
            // aligned output (rolls wrap around the frame, :42-43), kept window scaled by the mean-radius factor, soft demapper per axis (:48):
            // E_q[x_I] and the first maximum of q per axis, exactly what pass 1 derives from a materialised q
            float amp2[NLEV];
#pragma unroll
            for (int i = 0; i < NLEV; i++) amp2[i] = amp[i] * amp[i];
            const float nusc = nu_sc[run];
            for (int n = tid; n < (int)N; n += EPI_NT) {
                const bool kp = n >= EDGE && n < (int)N - EDGE - ms;
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int sp = (p - r) & 1;
                    int m = n + (p ? s1 : s0);
                    if (m >= (int)N) m -= (int)N;
                    if (m < 0) m += (int)N;
                    const float i2v = 0.5f / var[run * 2 + p];
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        float v = yr[(size_t)(sp * 2 + c) * N + m];
                        if (kp) v *= fac;
                        float qq[NLEV];
                        soft_demap<NLEV>(v, amp, amp2, i2v, nusc, qq);
                        float best = qq[0], e = amp[0] * qq[0];
                        int bi = 0;
#pragma unroll
                        for (int i = 1; i < NLEV; i++) {
                            if (c == 0) e = fmaf(amp[i], qq[i], e);
                            if (qq[i] > best) { best = qq[i]; bi = i; }
                        }
                        D[(size_t)(p * 2 + c) * N + n] = (int8_t)bi;
                        if (c == 0) E[(size_t)p * N + n] = e;
                    }
                }
            }
        }
        __syncthreads();                                        // (CMA: the workgroup's own global writes are visible to it after the barrier)
    }
}

}  // namespace vaeq

#include "vaeq_epilogue_lds.h"

extern "C" int64_t vaeq_dp_epilogue_ws_bytes(int32_t R, int64_t N)
{
    if (R < 0 || N < 0) return VAEQ_ERR_SHAPE;
    return (int64_t)R * N * (2 * 4 + 4);                        // E_q[x_I] floats [R][2][N] + hard decisions int8 [R][2][2][N]
}

extern "C" int vaeq_dp_epilogue(int32_t R, int64_t N, int32_t n_lev, int32_t batch_len, const float *q, const float *y, const void *tx_f16,
                                const float *amp, const float *var, const float *nu_sc, float *ser, int32_t *shift, int32_t *rflag,
                                void *workspace, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!q || !y || !tx_f16 || !amp || !var || !nu_sc || !ser || !shift || !rflag || !workspace) return VAEQ_ERR_NULL;
    if (R < 0 || N < 2 * vaeq::EDGE + vaeq::N_SHIFT || N > 0x3fffffff || batch_len < 0 || (batch_len > 0 && N % batch_len)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *wsE = reinterpret_cast<float *>(workspace);
    int8_t *wsD = reinterpret_cast<int8_t *>(wsE + (size_t)R * 2 * N);
    const __half *tx = reinterpret_cast<const __half *>(tx_f16);
    switch (n_lev) {
    case 2: hipLaunchKernelGGL(vaeq::dp_epilogue_kernel<2>, dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    case 4: hipLaunchKernelGGL(vaeq::dp_epilogue_kernel<4>, dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    case 8: hipLaunchKernelGGL(vaeq::dp_epilogue_kernel<8>, dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    default: return VAEQ_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_dp_epilogue_compact(int32_t R, int64_t N, int32_t n_lev, int32_t batch_len, const float *eq, const int8_t *dec, const float *y,
                                        const void *tx_f16, const float *amp, const float *var, const float *nu_sc, float *ser, int32_t *shift,
                                        int32_t *rflag, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!eq || !dec || !y || !tx_f16 || !amp || !var || !nu_sc || !ser || !shift || !rflag) return VAEQ_ERR_NULL;
    if (R < 0 || N < 2 * vaeq::EDGE + vaeq::N_SHIFT || N > 0x3fffffff || batch_len < 0 || (batch_len > 0 && N % batch_len)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __half *tx = reinterpret_cast<const __half *>(tx_f16);
    // dp_epilogue_compact_kernel (vaeq_epilogue_lds.h): both correlations on one staged TX tile, branch-free walks, TX levels resident in LDS when the
    // frame's packed rows fit beside the tiles (N <= ~60 000 symbols).  VAEQ_EPI_REREAD=1 keeps the re-reading dp_epilogue_kernel, VAEQ_EPI_NOTXC=1 the
    // new kernel without the TX level cache (A/B switches and cross-checks: all three agree bit for bit)
    const char *old_env = getenv("VAEQ_EPI_REREAD"), *notxc_env = getenv("VAEQ_EPI_NOTXC");
    if ((batch_len == 0 || batch_len >= 4) && !(old_env && old_env[0] == '1')) {
        const size_t dyn_txc = (size_t)4 * vaeq::nib_words((int)N) * sizeof(uint32_t);
        const bool txc = dyn_txc + sizeof(vaeq::Epi2Shared) <= 53 * 1024 && !(notxc_env && notxc_env[0] == '1');   // three workgroups per CU
        const size_t dyn = txc ? dyn_txc : 0;
#define VAEQ_EPI2(NL)                                                                                                                     \
    {                                                                                                                                     \
        auto k = txc ? vaeq::dp_epilogue_compact_kernel<NL, true> : vaeq::dp_epilogue_compact_kernel<NL, false>;                          \
        hipLaunchKernelGGL(k, dim3(R), dim3(vaeq::EPI_NT), dyn, st, (int)N, batch_len, eq, dec, y, tx, amp, var, nu_sc, ser, shift, rflag); \
    }
        switch (n_lev) {
        case 2: VAEQ_EPI2(2) break;
        case 4: VAEQ_EPI2(4) break;
        case 8: VAEQ_EPI2(8) break;
        default: return VAEQ_ERR_SHAPE;
        }
#undef VAEQ_EPI2
        return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
    }
    float *wsE = const_cast<float *>(eq);
    int8_t *wsD = const_cast<int8_t *>(dec);
    const float *q = nullptr;
    switch (n_lev) {
    case 2: hipLaunchKernelGGL(vaeq::dp_epilogue_kernel<2>, dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    case 4: hipLaunchKernelGGL(vaeq::dp_epilogue_kernel<4>, dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    case 8: hipLaunchKernelGGL(vaeq::dp_epilogue_kernel<8>, dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    default: return VAEQ_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

// The constant-modulus baselines' two-stage epilogue in one launch (func_CMA_DP_MQAM_shaping.py:39-52 after the phase estimation; identical in the
// CMAbatch / CMAflex modules): y[R][2][2][N] = CPE output (already cut to [10:-10]), tx_f16[R][2][2][N] the TX reference cut likewise;
// ser[R][4] = constellation SER of both polarisations, then soft-demapper SER; shift[R][2][2] / rflag[R][2]: index 0 = soft-demapper stage, 1 =
// constellation stage; workspace: vaeq_dp_epilogue_ws_bytes(R, N).
extern "C" int vaeq_cma_epilogue(int32_t R, int64_t N, int32_t n_lev, const float *y, const void *tx_f16, const float *amp, const float *var,
                                 const float *nu_sc, float *ser, int32_t *shift, int32_t *rflag, void *workspace, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!y || !tx_f16 || !amp || !var || !nu_sc || !ser || !shift || !rflag || !workspace) return VAEQ_ERR_NULL;
    if (R < 0 || N < 2 * vaeq::EDGE + vaeq::N_SHIFT || N > 0x3fffffff) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *wsE = reinterpret_cast<float *>(workspace);
    int8_t *wsD = reinterpret_cast<int8_t *>(wsE + (size_t)R * 2 * N);
    const __half *tx = reinterpret_cast<const __half *>(tx_f16);
    const float *q = nullptr;
    const int batch_len = 0;
    switch (n_lev) {
    case 2: hipLaunchKernelGGL((vaeq::dp_epilogue_kernel<2, true>), dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    case 4: hipLaunchKernelGGL((vaeq::dp_epilogue_kernel<4, true>), dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    case 8: hipLaunchKernelGGL((vaeq::dp_epilogue_kernel<8, true>), dim3(R), dim3(vaeq::EPI_NT), 0, st, N, batch_len, q, y, tx, amp, var, nu_sc, ser, shift, rflag, wsE, wsD); break;
    default: return VAEQ_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}
