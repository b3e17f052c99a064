// vaeq_epilogue_lds.h -- the per-frame DP epilogue (R12 / f2) fed by the training kernel's compact outputs: dp_epilogue_compact_kernel.
// (included by vaeq_epilogue.hip after dp_epilogue_kernel, whose helpers -- KeepWalk4, epi_ld_*, the constants -- it shares)
//
// Same results as dp_epilogue_kernel in its q == nullptr mode (shared_funcs.py:188-338 and the roll / cut logic of
// func_VAELE_DP_MQAM_shaping.py:68-89, see the head of vaeq_epilogue.hip), bit for bit: same tiles, same summation orders, same decisions
// (tests: test_lds_resident_epilogue_equals_rereading_epilogue, test_compact_outputs_equal_what_the_epilogue_derives_from_q).  What changes:
//   * ONE pass over the symbol axis stages each TX tile once and runs BOTH 21-lag correlations on it (E_q[x_I] and y[:, 0, :] against TX); the
//     next tile's global loads are issued into registers before the current tile's 484 packed FMAs per thread, so the pass no longer alternates
//     between waiting for HBM and computing;
//   * TXC: while staging, TX is reduced to what the SER walks need of it -- the level index rint(scale t + scale) (:198) -- and kept for the
//     whole frame in LDS, two symbols per byte (4 rows x N / 2 bytes = 20 KB): TX is read from HBM twice instead of five times (the radius
//     walk takes |TX| from the fp16 reference itself);
//   * the SER walks are bound by instruction issue, not by memory (measured: phase stamps; cutting the traffic alone changed nothing): the
//     re-reading kernel spends ~120 vector + scalar instructions per symbol and polarisation on the eight hypotheses (4 rotations x IQ flip).
//     Here a symbol's eight error indicators are ONE byte of a 4096-entry table in LDS indexed by (TX level I, TX level Q, decision level I,
//     decision level Q) -- for the constellation path the decision levels are the intervals the scaled output falls into (seven compares per
//     axis; the negated rotations follow from the thresholds' symmetry, with an exact per-symbol fallback) -- whose bits are spread over packed
//     byte counters with one multiply and one mask per four hypotheses; members of a group that are not kept add zero (no control flow).
// One 256-thread workgroup per run; 29 KB static LDS (incl. the 4 KB hypothesis table) + (TXC) 4 x (N / 8 + 5) x 4 bytes dynamic: three per CU.
#pragma once

namespace vaeq {

struct Epi2Shared {
    float corr[2][2][2][2][N_SHIFT];          // [path][c][b][a][lag]
    int shift[2][2];                          // [path][pol]
    int r[2];
    int cnt[16];
    int kept;
    float lo[8], hi[8];
    float red[64];
    int sym;                                  // the decision thresholds are symmetric about zero (they are whenever the levels are)
    unsigned char lut[4096];                  // [a1][a0][dQ][dI] (3 bits each): bit 2k + f = hypothesis (rotation k, IQ flip f) is in error
    union {
        struct {
            float2 txs[2 * EPI_TILE];                 // TX tile: [symbol][a] = (I, Q) of polarisation a
            alignas(16) float es[2][2][ES_LEN];       // [path][b]: equaliser-side tiles with a 10-symbol halo on both sides
        } t;
        float2 part[N_CHUNK][2][2][N_SHIFT];          // per-chunk partial correlations of one path (after the last tile)
    } u;
};

constexpr int NIB_FRONT = 16, NIB_BACK = 24;  // pad nibbles around a row
__host__ __device__ inline int nib_words(int N) { return (N + NIB_FRONT + NIB_BACK + 7) / 8; }   // u32 per row

// four consecutive nibbles starting at symbol m (any alignment) of a packed row: symbol m sits at bits 4 ((m + 16) & 7) of word (m + 16) >> 3
__device__ __forceinline__ uint32_t nib4(const uint32_t *row, int m)
{
    const int k = m + NIB_FRONT;
    const uint64_t w = (uint64_t)row[k >> 3] | ((uint64_t)row[(k >> 3) + 1] << 32);
    return (uint32_t)(w >> (4 * (k & 7))) & 0xffffu;
}

// both correlations on the tiles staged in sh.u.t: thread = (chunk of 22 symbols, b, a, half of the lags); acc[path][lag]
__device__ __forceinline__ void epi2_tile_corr(const Epi2Shared &sh, int chunk, int b, int a, int h, v2f (&acc)[2][N_LAGH])
{
    const float2 *tp = sh.u.t.txs + 2 * (chunk * EPI_CH) + a;
#pragma unroll
    for (int path = 0; path < 2; path++) {
        const float *eb = sh.u.t.es[path][b] + chunk * EPI_CH + HALF_SHIFT * (1 - h);
        v2f W[16];
#pragma unroll
        for (int k = 0; k < 16; k++) W[k] = lds2(eb + 2 * k);
#pragma unroll
        for (int s = 0; s < EPI_CH; s++) {
            const v2f tv = lds2(tp + 2 * s);
#pragma unroll
            for (int i = 0; i < N_LAGH; i++) {
                const int e = s + HALF_SHIFT - i;
                acc[path][i] += tv * ((e & 1) ? W[e >> 1].y : W[e >> 1].x);
            }
        }
    }
}

// keep mask of the group of four symbols n0 .. n0 + 3 without control flow (KeepWalk4::keep4 with selects; batch_len >= 4 or 0)
__device__ __forceinline__ void keep4_sel(const KeepWalk4 &kw, int n0, bool (&kp)[4])
{
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int n = n0 + e;
        if (kw.batch_len <= 0) kp[e] = n >= EDGE && n < kw.N - EDGE - kw.ms;
        else {
            const bool wrap = kw.j + e >= kw.batch_len;
            const int je = kw.j + e - (wrap ? kw.batch_len : 0), me = kw.mb + (wrap ? 1 : 0);
            const int k = me * kw.Lk + je;
            kp[e] = n < kw.N && je < kw.Lk && k >= EDGE && k < kw.K - EDGE - kw.ms;
        }
    }
}

// group loaders of the walks with the access form as a compile-time choice: W = the group and its shifted partners lie inside the row (every group
// but the first three and the last four of a frame) -> plain vector loads, nothing conditional, so the compiler issues all loads of an iteration
// back to back and waits once; else the kept members one by one.  (With the form as a run-time flag every load ends up in a branch of its own,
// followed by its own s_waitcnt vmcnt(0): four to eight serialised HBM round trips per iteration -- that, not instruction count or traffic, is
// what the re-reading kernel's walks spend their time on.)
// (the member-by-member forms are rare -- seven groups per frame -- and deliberately NOT inlined: with them, and with the exact fallback of the
//  constellation decisions below, unrolled into every walk the kernel grew to 20 000 instructions, more than twice the instruction cache)
// (results by value: an array handed to a non-inlined function by reference would live in scratch for the fast path, too)
__device__ __attribute__((noinline)) float4 ld_y4_slow(const float *row, int m0, int kmask)
{
    float4 o;
    o.x = kmask & 1 ? row[m0] : 0.f; o.y = kmask & 2 ? row[m0 + 1] : 0.f; o.z = kmask & 4 ? row[m0 + 2] : 0.f; o.w = kmask & 8 ? row[m0 + 3] : 0.f;
    return o;
}
__device__ __attribute__((noinline)) uint32_t ld_d4_slow(const int8_t *row, int m0, int kmask)          // four decisions, one per byte
{
    uint32_t w = 0u;
    for (int e = 0; e < 4; e++)
        if ((kmask >> e) & 1) w |= ((uint32_t)row[m0 + e] & 7u) << (8 * e);
    return w;
}
__device__ __attribute__((noinline)) float4 ld_tx4_slow(const __half *row, int n0, int kmask)
{
    float4 o;
    o.x = kmask & 1 ? __half2float(row[n0]) : 0.f; o.y = kmask & 2 ? __half2float(row[n0 + 1]) : 0.f;
    o.z = kmask & 4 ? __half2float(row[n0 + 2]) : 0.f; o.w = kmask & 8 ? __half2float(row[n0 + 3]) : 0.f;
    return o;
}
__device__ __forceinline__ int kmask4(const bool (&kp)[4]) { return (int)kp[0] | ((int)kp[1] << 1) | ((int)kp[2] << 2) | ((int)kp[3] << 3); }

template <bool W>
__device__ __forceinline__ void ld_y4(const float *row, int m0, const bool (&kp)[4], float (&out)[4])
{
    if constexpr (W) {
        const f4u v = *reinterpret_cast<const f4u *>(row + m0);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
        const float4 v = ld_y4_slow(row, m0, kmask4(kp));
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    }
}
template <bool W>
__device__ __forceinline__ void ld_d4(const int8_t *row, int m0, const bool (&kp)[4], int (&out)[4])
{
    if constexpr (W) {
        uint32_t w;
        __builtin_memcpy(&w, row + m0, 4);
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = (int)((w >> (8 * e)) & 7u);
    } else {
        const uint32_t w = ld_d4_slow(row, m0, kmask4(kp));
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = (int)((w >> (8 * e)) & 7u);
    }
}
template <bool W>
__device__ __forceinline__ void ld_tx4(const __half *row, int n0, const bool (&kp)[4], float (&out)[4])
{
    if constexpr (W) {
        const uint2 w = *reinterpret_cast<const uint2 *>(row + n0);
        const __half2 h01 = *reinterpret_cast<const __half2 *>(&w.x), h23 = *reinterpret_cast<const __half2 *>(&w.y);
        out[0] = __low2float(h01); out[1] = __high2float(h01); out[2] = __low2float(h23); out[3] = __high2float(h23);
    } else {
        const float4 v = ld_tx4_slow(row, n0, kmask4(kp));
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    }
}

// constellation decisions of one symbol when the symmetric short cut does not apply (an output exactly on a threshold, asymmetric thresholds): the
// intervals of yi, -yi, yq, -yq each by their own compares (hi: the S thresholds, in LDS) -> error bits of the eight hypotheses (:245-287)
__device__ __attribute__((noinline)) uint32_t epi_eb_generic(float yi, float yq, const float *hi, int S, int dI, int dQ)
{
    int A = 0, Am = 0, Bq = 0, Bm = 0;
    for (int i = 0; i < S; i++) {
        const float t = hi[i];
        A += (int)(yi >= t); Am += (int)(-yi >= t); Bq += (int)(yq >= t); Bm += (int)(-yq >= t);
    }
    const int X[4] = {A, Am, Bm, Bq}, Y[4] = {Bq, Bm, A, Am}, dQf = S - dQ;
    uint32_t eb = 0u;
    for (int k = 0; k < 4; k++) {
        eb |= (uint32_t)((X[k] != dI) | (Y[k] != dQ)) << (2 * k);
        eb |= (uint32_t)((X[k] != dI) | (Y[k] != dQf)) << (2 * k + 1);
    }
    return eb;
}

// what a thread holds of the NEXT tile while the current one is being correlated
struct EpiTileRegs {
    uint2 tx[4];                              // TX group (4 symbols x 4 rows, fp16) of thread tid < EPI_TILE / 4
    float4 es[3];                             // equaliser-side groups tid, tid + 256, tid + 512 of the 4 x EG
};

template <int NLEV, bool TXC>
__global__ __launch_bounds__(EPI_NT, 3) void dp_epilogue_compact_kernel(int N, int batch_len, const float *__restrict__ eq, const int8_t *__restrict__ dec,
                                                                                    const float *__restrict__ y, const __half *__restrict__ txg,
                                                                                    const float *__restrict__ amp_g, const float *__restrict__ var,
                                                                                    const float *__restrict__ nu_sc, float *__restrict__ ser,
                                                                                    int32_t *__restrict__ shift_out, int32_t *__restrict__ r_out)
{
#ifdef VAEQ_EPI_STAMPS       // profiling build (tools/probe_epi_phases.py): shader-clock stamps of the workgroup of run gridDim.x / 2, left in eq[0][0..]
    long long stp[8];
    int nst = 0;
#define EPI_STAMP() do { stp[nst++] = __builtin_readcyclecounter(); } while (0)
#else
#define EPI_STAMP() do { } while (0)
#endif
    __shared__ Epi2Shared sh;
    extern __shared__ uint32_t nibs[];                          // TXC: [tx rows 0..3][NWD] level nibbles of the whole frame
    const int run = blockIdx.x, tid = threadIdx.x, NWD = nib_words(N);
    const float *E = eq + (size_t)run * 2 * N, *yr = y + (size_t)run * 4 * N;
    const int8_t *D = dec + (size_t)run * 4 * N;
    const __half *tx = txg + (size_t)run * 4 * N;              // [a][c][n]
    uint32_t *txn = nibs;
    constexpr int S = NLEV - 1;
    const float scale = 0.5f * S;
    if (tid == 0) {
        const float thr_scale = 1.0f + 2.0f * nu_sc[run] * var[run * 2 + 0];      // :234
        for (int i = 0; i < NLEV; i++) {
            sh.lo[i] = i == 0 ? -INFINITY : thr_scale * (amp_g[i - 1] + amp_g[i]) / 2;
            sh.hi[i] = i == NLEV - 1 ? INFINITY : thr_scale * (amp_g[i] + amp_g[i + 1]) / 2;
        }
        for (int i = 0; i < 16; i++) sh.cnt[i] = 0;
        sh.kept = 0;
    }
    if (TXC)
        for (int i = tid; i < 4 * NWD; i += EPI_NT) nibs[i] = 0;
#pragma unroll 1
    for (int idx = tid; idx < 4096; idx += EPI_NT) {           // hypothesis table: decisions (a0, a1) under the four rotations (:201-217) against
        const int dI = idx & 7, dQ = (idx >> 3) & 7, a0 = (idx >> 6) & 7, a1 = idx >> 9, dQf = S - dQ;    // the TX levels (dI, dQ) and their IQ flip (:199)
        const int hI[4] = {a0, S - a0, S - a1, a1}, hQ[4] = {a1, S - a1, a0, S - a0};
        uint32_t eb = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            eb |= (uint32_t)((dI != hI[k]) | (dQ != hQ[k])) << (2 * k);
            eb |= (uint32_t)((dI != hI[k]) | (dQf != hQ[k])) << (2 * k + 1);
        }
        sh.lut[idx] = (unsigned char)eb;
    }
    __syncthreads();
    if (tid == 0) {
        int sym = 1;
        for (int i = 0; i < S; i++) sym &= (int)(sh.hi[i] == -sh.hi[S - 1 - i]);
        sh.sym = sym;
    }
    __syncthreads();
    EPI_STAMP();

    // ---- pass A: one walk over the symbol axis in tiles: TX staged once (TXC: + its levels packed into LDS for the whole frame), both correlations
    const int chunk = tid >> 3, b = (tid >> 2) & 1, a = (tid >> 1) & 1, h = tid & 1;
    const bool wide_tx = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(tx) & 7) == 0;
    constexpr int EG = (EPI_TILE + 2 * HALF_SHIFT) / 4;        // equaliser-side rows [t0 - 10 .. t0 + TILE + 10), indices mod N
    static_assert(EPI_TILE / 4 <= EPI_NT && 4 * EG <= 3 * EPI_NT, "one TX group and three equaliser-side groups per thread");
    auto load_tile = [&](int t0, EpiTileRegs &rg) {
        if (tid < EPI_TILE / 4) {                               // TX rows; symbols past the end contribute zeros
            const int n = t0 + 4 * tid;
            if (wide_tx && n + 3 < N) {
#pragma unroll
                for (int row = 0; row < 4; row++) rg.tx[row] = *reinterpret_cast<const uint2 *>(tx + (size_t)row * N + n);
            } else {
#pragma unroll
                for (int row = 0; row < 4; row++) {
                    unsigned short hv[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) hv[e] = n + e < N ? __half_as_ushort(tx[(size_t)row * N + n + e]) : (unsigned short)0;
                    rg.tx[row] = make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + k * EPI_NT;
            if (i < 4 * EG) {
                const int pb = i / EG, j = 4 * (i - pb * EG), m0 = t0 + j - HALF_SHIFT, path = pb >> 1, bb = pb & 1;
                const float *row = path == 0 ? E + (size_t)bb * N : yr + (size_t)bb * 2 * N;  // path 1: y[:, 0, :] = rows 0 and 2 of y[2][2][N] (:321)
                if (m0 >= 0 && m0 + 3 < N) {
                    const f4u w = *reinterpret_cast<const f4u *>(row + m0);
                    rg.es[k] = make_float4(w.x, w.y, w.z, w.w);
                } else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        int m = m0 + e;
                        while (m < 0) m += N;
                        while (m >= N) m -= N;
                        t[e] = row[m];
                    }
                    rg.es[k] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        }
    };
    auto store_tile = [&](int t0, const EpiTileRegs &rg) {
        if (tid < EPI_TILE / 4) {
            const int n = t0 + 4 * tid;
            float v[4][4];
#pragma unroll
            for (int row = 0; row < 4; row++) {
                const __half2 h01 = *reinterpret_cast<const __half2 *>(&rg.tx[row].x), h23 = *reinterpret_cast<const __half2 *>(&rg.tx[row].y);
                v[row][0] = __low2float(h01); v[row][1] = __high2float(h01); v[row][2] = __low2float(h23); v[row][3] = __high2float(h23);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sh.u.t.txs[2 * (4 * tid + e) + 0] = make_float2(v[0][e], v[1][e]);
                sh.u.t.txs[2 * (4 * tid + e) + 1] = make_float2(v[2][e], v[3][e]);
            }
            if (TXC && n < N) {                                 // level indices (:198), four per 16-bit store (n is a multiple of 4)
#pragma unroll
                for (int row = 0; row < 4; row++) {
                    uint32_t pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) pk |= (uint32_t)min(max((int)rintf(scale * v[row][e] + scale), 0), 15) << (4 * e);
                    reinterpret_cast<uint16_t *>(txn + row * NWD)[(n + NIB_FRONT) >> 2] = (uint16_t)pk;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + k * EPI_NT;
            if (i < 4 * EG) {
                const int pb = i / EG, j = 4 * (i - pb * EG);
                *reinterpret_cast<float4 *>(&sh.u.t.es[pb >> 1][pb & 1][j]) = rg.es[k];
            }
        }
    };
    v2f acc[2][N_LAGH];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int i = 0; i < N_LAGH; i++) acc[p][i] = v2f{0.f, 0.f};
    {
        EpiTileRegs rg;
        load_tile(0, rg);
        for (int t0 = 0; t0 < N; t0 += EPI_TILE) {
            store_tile(t0, rg);
            __syncthreads();
            if (t0 + EPI_TILE < N) load_tile(t0 + EPI_TILE, rg);   // in flight while this tile is correlated
            epi2_tile_corr(sh, chunk, b, a, h, acc);
            __syncthreads();
        }
    }
    EPI_STAMP();
#pragma unroll
    for (int path = 0; path < 2; path++) {                      // per path: partial sums -> fixed-order sum over chunks -> |corr|
#pragma unroll
        for (int i = 0; i < N_LAGH; i++)
            if (h == 0 || i > 0) sh.u.part[chunk][b][a][HALF_SHIFT * h + i] = make_float2(acc[path][i].x, acc[path][i].y);
        __syncthreads();
        if (tid < 2 * 2 * N_SHIFT) {
            const int l = tid % N_SHIFT, ba = tid / N_SHIFT, b2 = ba >> 1, a2 = ba & 1;
            v2f tot = {0.f, 0.f};
            for (int k = 0; k < N_CHUNK; k++) tot += lds2(&sh.u.part[k][b2][a2][l]);
            sh.corr[path][0][b2][a2][l] = fabsf(tot.x);
            sh.corr[path][1][b2][a2][l] = fabsf(tot.y);
        }
        __syncthreads();
    }
    if (tid < 2) {                                              // shared_funcs.py:303-314, one thread per path
        const int path = tid;
        float cm[2][2];
        int pick[2][2];
        for (int bb = 0; bb < 2; bb++)
            for (int aa = 0; aa < 2; aa++) {
                float best[2];
                int bi[2];
                for (int c = 0; c < 2; c++) {
                    best[c] = sh.corr[path][c][bb][aa][0];
                    bi[c] = 0;
                    for (int l = 1; l < N_SHIFT; l++)
                        if (sh.corr[path][c][bb][aa][l] > best[c]) { best[c] = sh.corr[path][c][bb][aa][l]; bi[c] = l; }
                }
                const int cw = best[1] > best[0] ? 1 : 0;       // torch.max over (I, Q): first maximum wins ties
                cm[bb][aa] = best[cw];
                pick[bb][aa] = bi[cw];
            }
        const bool straight = (cm[0][0] + cm[1][1]) >= (cm[0][1] + cm[1][0]);
        sh.shift[path][0] = HALF_SHIFT - (straight ? pick[0][0] : pick[0][1]);
        sh.shift[path][1] = HALF_SHIFT - (straight ? pick[1][1] : pick[1][0]);
        sh.r[path] = straight ? 0 : 1;
        shift_out[(size_t)run * 4 + path * 2 + 0] = sh.shift[path][0];
        shift_out[(size_t)run * 4 + path * 2 + 1] = sh.shift[path][1];
        r_out[(size_t)run * 2 + path] = sh.r[path];
    }
    __syncthreads();
    EPI_STAMP();

    const bool row_wide = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(tx) & 7) == 0 && (reinterpret_cast<uintptr_t>(D) & 3) == 0;
    const int NG = (N + 3) >> 2;
    const bool sym = sh.sym != 0;
    // interior groups (every member's partner index n + shift inside the row: n0 >= 12, n0 + 14 <= N) take the unconditional vector loads
    const int gA = row_wide ? min(3, NG) : NG, gB = row_wide ? max(gA, (N - 14) / 4 + 1) : NG;
    // walk<R>(load, comp): every group g = tid, tid + 256, ... with a kept member, in that order, EPI_U groups per round: first the loads of all
    // EPI_U groups (load(tag, n0, kp, R&), tag = true_type for interior groups), then their arithmetic (comp(n0, kp, R)) -- a walk is bound by
    // the HBM round trip per round (2-3 us under load), so what counts is how many loads a wave has in flight per round trip
#ifndef VAEQ_EPI_U
#define VAEQ_EPI_U 2
#endif
    constexpr int EPI_U = VAEQ_EPI_U;
    auto walk = [&](int s0, int ms, auto rtag, auto load, auto comp) {
        using R = decltype(rtag);
        KeepWalk4 kw(4 * tid, N, batch_len, s0, ms);
        for (int g0 = tid; g0 < NG; g0 += EPI_U * EPI_NT) {
            R rg[EPI_U];
            bool kp[EPI_U][4], any[EPI_U];
#pragma unroll
            for (int u = 0; u < EPI_U; u++) {
                const int g = g0 + u * EPI_NT;
                keep4_sel(kw, 4 * g, kp[u]);                    // (groups past the end: n < N fails for every member)
                any[u] = kp[u][0] || kp[u][1] || kp[u][2] || kp[u][3];
                kw.next();
            }
#pragma unroll
            for (int u = 0; u < EPI_U; u++) {
                const int g = g0 + u * EPI_NT;
                if (any[u]) {
                    if (g >= gA && g < gB) load(std::true_type{}, 4 * g, kp[u], rg[u]);
                    else load(std::false_type{}, 4 * g, kp[u], rg[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < EPI_U; u++)
                if (any[u]) comp(4 * (g0 + u * EPI_NT), kp[u], rg[u]);
        }
    };
    struct RadRegs { float yi[2][4], yq[2][4], ti[2][4], tq[2][4]; };
    struct SerRegs { uint32_t li[2], lq[2]; float u0[2][4], u1[2][4]; };       // TX levels packed 4 per word; u0 / u1: decisions (path 0) or outputs (path 1)
#pragma unroll
    for (int path = 0; path < 2; path++) {                      // 0: soft-demapper path, 1: constellation path on y
        const int s0 = sh.shift[path][0], s1 = sh.shift[path][1], r = sh.r[path];
        const int ms = max(abs(s0), abs(s1));
        float fac = 1.0f;
        if (path == 1) {                                        // mean radius of TX over mean radius of the aligned output (:242)
            float st = 0.f, sy = 0.f;                           // (|TX| from the fp16 reference itself: its only re-read)
            walk(s0, ms, RadRegs{},
                 [&](auto wt, int n0, const bool (&kp)[4], RadRegs &q) {
                     constexpr bool W = decltype(wt)::value;
#pragma unroll
                     for (int p = 0; p < 2; p++) {
                         const int sp = (p - r) & 1, m0 = n0 + (p ? s1 : s0);
                         ld_y4<W>(yr + (size_t)(sp * 2 + 0) * N, m0, kp, q.yi[p]);
                         ld_y4<W>(yr + (size_t)(sp * 2 + 1) * N, m0, kp, q.yq[p]);
                         ld_tx4<W>(tx + (size_t)(p * 2 + 0) * N, n0, kp, q.ti[p]);
                         ld_tx4<W>(tx + (size_t)(p * 2 + 1) * N, n0, kp, q.tq[p]);
                     }
                 },
                 [&](int, const bool (&kp)[4], const RadRegs &q) {
#pragma unroll
                     for (int p = 0; p < 2; p++) {
#pragma unroll
                         for (int e = 0; e < 4; e++) {          // (adding 0 for the members that are not kept leaves the sums bit-identical)
                             st += kp[e] ? sqrtf(fmaf(q.ti[p][e], q.ti[p][e], q.tq[p][e] * q.tq[p][e])) : 0.f;
                             sy += kp[e] ? sqrtf(fmaf(q.yi[p][e], q.yi[p][e], q.yq[p][e] * q.yq[p][e])) : 0.f;
                         }
                     }
                 });
            block_reduce3<EPI_NT>(st, sy, 0.f, sh.red);
            fac = sh.red[0] / sh.red[1];
            __syncthreads();
            EPI_STAMP();
        }
        // Error indicators of the eight hypotheses (4 rotations x IQ flip) of one symbol = one byte of sh.lut, indexed by the TX levels of both axes and
        // the two decision levels (soft demapper: argmax q per axis; constellation: the decision interval the scaled output falls into); the byte's
        // bits are spread over the bytes of two words and added to packed counters (flushed before a byte can overflow).
        int cnt[16];
#pragma unroll
        for (int i = 0; i < 16; i++) cnt[i] = 0;
        uint32_t accL[2] = {0u, 0u}, accH[2] = {0u, 0u};
        auto flush = [&]() {
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int hh = 0; hh < 4; hh++) {
                    cnt[hh * 2 + p] += (int)((accL[p] >> (8 * hh)) & 0xffu);
                    cnt[(4 + hh) * 2 + p] += (int)((accH[p] >> (8 * hh)) & 0xffu);
                }
                accL[p] = accH[p] = 0u;
            }
        };
        auto add_err = [&](int p, uint32_t eb) {                 // eb: bit h = hypothesis h is in error for this symbol
            accL[p] += ((eb & 15u) * 0x00204081u) & 0x01010101u;
            accH[p] += ((eb >> 4) * 0x00204081u) & 0x01010101u;
        };
        float thr[NLEV > 1 ? NLEV - 1 : 1];
#pragma unroll
        for (int i = 0; i < S; i++) thr[i] = sh.hi[i];
        int kept = 0, since = 0;
        walk(s0, ms, SerRegs{},
             [&](auto wt, int n0, const bool (&kp)[4], SerRegs &q) {
                 constexpr bool W = decltype(wt)::value;
#pragma unroll
                 for (int p = 0; p < 2; p++) {
                     const int sp = (p - r) & 1;                 // roll(r, 0): row p comes from row p - r  (:71)
                     const int m0 = n0 + (p ? s1 : s0);          // roll(-shift): out[n] = in[n + shift]     (:72)
                     if (TXC) {                                  // TX level indices (:198): from the frame's nibbles in LDS ...
                         q.li[p] = nib4(txn + (p * 2 + 0) * NWD, n0);
                         q.lq[p] = nib4(txn + (p * 2 + 1) * NWD, n0);
                     } else {                                    // ... or from HBM
                         float t0[4], t1[4];
                         ld_tx4<W>(tx + (size_t)(p * 2 + 0) * N, n0, kp, t0);
                         ld_tx4<W>(tx + (size_t)(p * 2 + 1) * N, n0, kp, t1);
                         q.li[p] = q.lq[p] = 0u;
#pragma unroll
                         for (int e = 0; e < 4; e++) {
                             q.li[p] |= (uint32_t)min(max((int)rintf(scale * t0[e] + scale), 0), 15) << (4 * e);
                             q.lq[p] |= (uint32_t)min(max((int)rintf(scale * t1[e] + scale), 0), 15) << (4 * e);
                         }
                     }
                     if (path == 0) {
                         int d0[4], d1[4];
                         ld_d4<W>(D + (size_t)(sp * 2 + 0) * N, m0, kp, d0);
                         ld_d4<W>(D + (size_t)(sp * 2 + 1) * N, m0, kp, d1);
#pragma unroll
                         for (int e = 0; e < 4; e++) { q.u0[p][e] = __int_as_float(d0[e]); q.u1[p][e] = __int_as_float(d1[e]); }
                     } else {
                         ld_y4<W>(yr + (size_t)(sp * 2 + 0) * N, m0, kp, q.u0[p]);
                         ld_y4<W>(yr + (size_t)(sp * 2 + 1) * N, m0, kp, q.u1[p]);
                     }
                 }
             },
             [&](int, const bool (&kp)[4], const SerRegs &q) {
                 kept += (int)kp[0] + (int)kp[1] + (int)kp[2] + (int)kp[3];
#pragma unroll
                 for (int p = 0; p < 2; p++) {
#pragma unroll
                     for (int e = 0; e < 4; e++) {
                         const int lI = (int)((q.li[p] >> (4 * e)) & 15u), lQ = (int)((q.lq[p] >> (4 * e)) & 15u);
                         uint32_t eb;
                         if (path == 0) {
                             // decisions under rotation by 0, pi, pi/2, 3pi/2 (:201-217) against the TX levels and their IQ flip (:199): a TX level outside
                             // 0 .. NLEV - 1 matches no decision under any hypothesis
                             eb = sh.lut[(lI & 7) | ((lQ & 7) << 3) | (__float_as_int(q.u0[p][e]) << 6) | (__float_as_int(q.u1[p][e]) << 9)];
                             if ((lI | lQ) > S) eb = 0xffu;
                         } else {
                             // d_vec0[lev] <= v < d_vec1[lev] (:267-287) <=> lev == the interval v falls into = #{i : v >= hi[i]}.  The four rotations
                             // (:245-262) need the intervals of yi, -yi, yq, -yq: with thresholds symmetric about zero that of -v is S - interval(v) unless
                             // v sits exactly on a threshold -- then, and for asymmetric thresholds, -v is quantised by itself
                             const float yi = q.u0[p][e] * fac, yq = q.u1[p][e] * fac;
                             int A = 0, Bq = 0;
                             bool on = false;                   // an output exactly ON a threshold (strictly ascending thresholds: yi == hi[A - 1] <=> yi equals one
                                                                // of them) -- found by register compares: as `A > 0 && yi == sh.hi[A - 1]` each test was an LDS read in
                                                                // an exec-masked branch of its own behind its own s_waitcnt (eight per group of four symbols)
#pragma unroll
                             for (int i = 0; i < S; i++) {
                                 A += (int)(yi >= thr[i]); Bq += (int)(yq >= thr[i]);
                                 on |= (yi == thr[i]) | (yq == thr[i]);
                             }
                             const bool odd = !sym || on;
                             const int dI = min(lI, S), dQ = min(lQ, S);                      // (a level outside the range takes the nearest interval, :270)
                             eb = sh.lut[dI | (dQ << 3) | (A << 6) | (Bq << 9)];
                             if (odd) eb = epi_eb_generic(yi, yq, sh.hi, S, dI, dQ);
                             if (!(fabsf(yi) < INFINITY) || !(fabsf(yq) < INFINITY)) eb = 0xffu;  // a non-finite output (diverged run): an error under every hypothesis
                         }
                         add_err(p, kp[e] ? eb : 0u);
                     }
                 }
                 if (++since == 63) { flush(); since = 0; }
             });
        flush();
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
        __syncthreads();
        if (tid < 16) sh.cnt[tid] = 0;
        if (tid == 0) sh.kept = 0;
        __syncthreads();
        EPI_STAMP();
    }
#ifdef VAEQ_EPI_STAMPS       // stamps: start | pass A done | shifts done | SER q done | radius done | SER y done
    if (tid == 0 && run == (int)gridDim.x / 2)
        for (int i = 0; i + 1 < nst; i++) const_cast<float *>(eq)[i] = (float)(stp[i + 1] - stp[i]);
#endif
#undef EPI_STAMP
}

}  // namespace vaeq
