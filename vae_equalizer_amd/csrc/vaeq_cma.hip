// vaeq_cma.hip -- SURVEY row f4: the constant-modulus baselines of the DP scripts and their carrier phase estimation.
//
//   CMA (shared_funcs.py:341-383), CMAbatch (:385-433), CMAflex (:435-488): 2x2 real-coefficient butterfly FIR at 1 sample per
//   symbol output, error e = R - |out|^2, stochastic-gradient tap update after every symbol (CMA) or, with the increments of the
//   last `batchlen` symbols, whenever k % symb_step == 0 and k >= batchlen (CMAflex; CMAbatch = symb_step = batchlen).
//   The algorithms are sequential in the symbol index by construction; the parallelism is over taps (lanes) and runs (waves):
//   one 64-lane wave per run, lane t owns tap t of all 8 real filter rows h[out pol][in pol][re/im][t] (M <= 64).
//   Reference quirks kept: the frame is scaled by 1 / mean|y|^2 over the ZERO-PADDED length (a power, :350-351), and symbol j lands
//   at index k = (mh + sps j) / sps - mh, negative for the first symbols, i.e. wrapped to the end of out / e (:357).
//   CPE (:139-186): Viterbi-Viterbi 4th-power estimate, 501-symbol zero-padded moving average, atan2 / 4, unwrapping of the pi/2
//   jumps (the correction of sample n is the signed count of jumps before it), de-rotation; one workgroup per run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vaeq.h"
#include "vaeq_common.h"
#include "vaeq_wave.h"

#ifndef CMA_WAVES
#define CMA_WAVES 6
#endif
#ifndef CMA_UNROLL
#define CMA_UNROLL 2
#endif

namespace vaeq {

// HALF (M <= 32): lanes 0..31 hold the taps of output polarisation 0, lanes 32..63 those of output 1 -- a lane keeps 4 instead of 8 coefficients, a
// symbol costs two half-wave sums instead of four wave sums and half the FMAs of the filter and of the update.  !HALF: lane = tap of both outputs (M <= 64).
// STAGE (batch modes, when it fits): the scaled samples of the last sps (batchlen + 1) + M padded positions live in an LDS ring of xmask + 1 float4
// (both polarisations, re / im): the filter window and every term of the batch update are ONE 16-byte LDS read instead of four global loads with
// bounds checks and 64-bit address arithmetic (42 -> 13 instructions per update term).
template <bool HALF, bool STAGE>
__global__ __launch_bounds__(64, CMA_WAVES) void cma_kernel(int N, int sps, int M, int mode, int batchlen, int symb_step, int xmask,
                                                 const float *__restrict__ rx, float Rc, float *__restrict__ h, const float *__restrict__ lr,
                                                 float *__restrict__ out, float *__restrict__ eout)
{
    extern __shared__ float4 ring4[];                          // mode 1: [batchlen][8] floats = out[2][2], e[2], -, - ; STAGE: then xs[xmask + 1] float4
    float *ring = reinterpret_cast<float *>(ring4);
    float4 *xs = ring4 + 2 * (size_t)batchlen;
    constexpr int NO = HALF ? 1 : 2;                           // output polarisations a lane works for
    const int run = blockIdx.x, lane = threadIdx.x;
    const int tl = HALF ? (lane & 31) : lane, ob = HALF ? (lane >> 5) : 0;     // tap of this lane; its (first) output polarisation
    const int mh = M / 2, Lp = N + 2 * mh, K = N / sps;
    const float *x = rx + (size_t)run * 4 * N;                 // [pol][re/im][N]
    float pw = 0.f;
    for (int i = lane; i < 4 * N; i += 64) pw = fmaf(x[i], x[i], pw);
    const float inv = 1.0f / (wave_sum(pw) / (float)(2 * Lp));
    const bool tap = tl < M;
    float hr[NO][2], hi[NO][2];                                // [out pol (- ob)][in pol] at tap = tl
    float *hrun = h + (size_t)run * 8 * M;
#pragma unroll
    for (int o = 0; o < NO; o++)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            hr[o][p] = tap ? hrun[(((ob + o) * 2 + p) * 2 + 0) * M + tl] : 0.f;
            hi[o][p] = tap ? hrun[(((ob + o) * 2 + p) * 2 + 1) * M + tl] : 0.f;
        }
    const float two_lr = 2.0f * lr[run];
    float *orun = out + (size_t)run * 4 * K, *erun = eout ? eout + (size_t)run * 2 * K : nullptr;
    auto window_raw = [&](int i0, float (&yr)[2], float (&yi)[2]) {   // padded sample i0 + tap of both polarisations, unscaled
        const int sx = i0 + tl - mh;
        const bool ok = tap && sx >= 0 && sx < N;
        const int sc = ok ? sx : 0;                            // (one unconditional load per row: no divergent branch on the chain)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const float a_ = x[(p * 2 + 0) * N + sc], b_ = x[(p * 2 + 1) * N + sc];
            yr[p] = ok ? a_ : 0.f;
            yi[p] = ok ? b_ : 0.f;
        }
    };
    // The symbol loop is one dependency chain per run; what must not sit on it is memory latency.  (i) The window of symbol j + 1 is fetched while
    // symbol j is computed (and scaled only when it becomes the current one).  (ii) Outputs are parked one symbol per lane and leave 64 symbols at a
    // time as coalesced rows: gfx9 counts stores in vmcnt, so a per-symbol store by lane 0 made every next window wait for the previous symbol's
    // writes to land (2.3 us per symbol).  (iii) The sums run on the vector ALU (DPP): a shuffle butterfly is six dependent LDS round trips each.
    const int J = (N + sps - 1) / sps;                         // symbols: mh + sps j < N + mh
    const int joff = mh - mh / sps;                            // kraw = j - joff: (mh + sps j) / sps - mh
    float keep[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float yr[2], yi[2];
    // STAGE: padded position u = sample index + mh; xs[u & xmask] = scaled sample (zero outside the frame); positions < `filled` are in the ring
    auto sample4 = [&](int u) {                                // scaled sample at padded position u (global memory)
        const int n = u - mh;
        const bool ok = n >= 0 && n < N;
        const int nc = ok ? n : 0;
        const float a0 = x[nc], a1 = x[N + nc], a2 = x[2 * N + nc], a3 = x[3 * N + nc];
        return ok ? make_float4(a0 * inv, a1 * inv, a2 * inv, a3 * inv) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if constexpr (STAGE) {
        for (int u = lane; u < M + sps; u += 64) xs[u & xmask] = sample4(u);       // windows of symbols 0 and 1
        __syncthreads();
        const float4 v = xs[tl & xmask];
        yr[0] = tap ? v.x : 0.f; yi[0] = tap ? v.y : 0.f; yr[1] = tap ? v.z : 0.f; yi[1] = tap ? v.w : 0.f;
    } else {
        window_raw(0, yr, yi);
#pragma unroll
        for (int p = 0; p < 2; p++) { yr[p] *= inv; yi[p] *= inv; }
    }
    for (int j = 0; j < J; j++) {
        const int i0 = sps * j, kraw = j - joff;
        float yrn[2], yin[2];
        float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (STAGE) {
            if (lane < sps) nx = sample4(M + sps * (j + 1) + lane);               // the samples symbol j + 2 adds; written to the ring after this symbol
            const float4 v = xs[(i0 + sps + tl) & xmask];                         // the next symbol's window (already in the ring): off the chain
            yrn[0] = tap ? v.x : 0.f; yin[0] = tap ? v.y : 0.f; yrn[1] = tap ? v.z : 0.f; yin[1] = tap ? v.w : 0.f;
        } else {
            window_raw(i0 + sps, yrn, yin);                    // (past the end: zeros)
        }
        float o_[2][2], e_[2];
        if constexpr (HALF) {
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                re = fmaf(yr[p], hr[0][p], re); re = fmaf(-yi[p], hi[0][p], re);
                im = fmaf(yr[p], hi[0][p], im); im = fmaf(yi[p], hr[0][p], im);
            }
            re += dpp_f<0xB1>(re); im += dpp_f<0xB1>(im);      // sums inside each 16-lane row ...
            re += dpp_f<0x4E>(re); im += dpp_f<0x4E>(im);
            re += dpp_f<0x141>(re); im += dpp_f<0x141>(im);
            re += dpp_f<0x140>(re); im += dpp_f<0x140>(im);
            const int br = __builtin_bit_cast(int, re), bi = __builtin_bit_cast(int, im);
            auto rl = [](int v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, l)); };
            o_[0][0] = rl(br, 0) + rl(br, 16); o_[0][1] = rl(bi, 0) + rl(bi, 16);     // ... rows 0 + 1 = output 0, rows 2 + 3 = output 1
            o_[1][0] = rl(br, 32) + rl(br, 48); o_[1][1] = rl(bi, 32) + rl(bi, 48);
        } else {
#pragma unroll
            for (int o = 0; o < 2; o++) {
                float re = 0.f, im = 0.f;
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    re = fmaf(yr[p], hr[o][p], re); re = fmaf(-yi[p], hi[o][p], re);
                    im = fmaf(yr[p], hi[o][p], im); im = fmaf(yi[p], hr[o][p], im);
                }
                o_[o][0] = wave_sum_dpp(re);
                o_[o][1] = wave_sum_dpp(im);
            }
        }
#pragma unroll
        for (int o = 0; o < 2; o++) e_[o] = Rc - o_[o][0] * o_[o][0] - o_[o][1] * o_[o][1];
        if (lane == (j & 63)) {
            keep[0] = o_[0][0]; keep[1] = o_[0][1]; keep[2] = o_[1][0]; keep[3] = o_[1][1]; keep[4] = e_[0]; keep[5] = e_[1];
        }
        if ((j & 63) == 63 || j == J - 1) {                    // uniform: flush the parked symbols (j & ~63) .. j
            const int jl = (j & ~63) + lane;
            if (jl <= j) {
                const int kr = jl - joff, kl = kr < 0 ? kr + K : kr;
                orun[0 * K + kl] = keep[0]; orun[1 * K + kl] = keep[1]; orun[2 * K + kl] = keep[2]; orun[3 * K + kl] = keep[3];
                if (erun) { erun[kl * 2 + 0] = keep[4]; erun[kl * 2 + 1] = keep[5]; }
            }
        }
        if (mode == 0) {                                       // :371-381
#pragma unroll
            for (int o = 0; o < NO; o++) {
                const float oR = HALF ? (ob ? o_[1][0] : o_[0][0]) : o_[o][0], oI = HALF ? (ob ? o_[1][1] : o_[0][1]) : o_[o][1];
                const float ge = two_lr * (HALF ? (ob ? e_[1] : e_[0]) : e_[o]);
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    hr[o][p] += ge * (oR * yr[p] + oI * yi[p]);
                    hi[o][p] += ge * (oI * yr[p] - oR * yi[p]);
                }
            }
        } else if (kraw >= 0) {
            if (kraw >= batchlen && kraw % symb_step == 0) {   // :421 / :475: the last `batchlen` symbols, kraw itself excluded
                float ar[NO][2], ai[NO][2];
#pragma unroll
                for (int o = 0; o < NO; o++) ar[o][0] = ar[o][1] = ai[o][0] = ai[o][1] = 0.f;
                // the window of symbol kk starts at sample sps (kk + joff): no look-up in the ring, so the loads of two symbols are in flight at once;
                // ring slot and sample index advance incrementally (a run-time modulo per term cost as much as the term's arithmetic)
                int slot_i = kraw % batchlen;                  // (kraw - batchlen) mod batchlen
                if constexpr (STAGE) {
                    int u = sps * (kraw - batchlen + joff) + tl;
#pragma unroll 4
                    for (int kk = kraw - batchlen; kk < kraw; kk++) {
                        const float *s = ring + (size_t)slot_i * 8;
                        const float4 w = xs[u & xmask];
                        const float wr[2] = {w.x, w.z}, wi[2] = {w.y, w.w};
#pragma unroll
                        for (int o = 0; o < NO; o++) {
                            const float sR = HALF ? s[ob * 2] : s[o * 2], sI = HALF ? s[ob * 2 + 1] : s[o * 2 + 1], sE = HALF ? s[4 + ob] : s[4 + o];
#pragma unroll
                            for (int p = 0; p < 2; p++) {
                                ar[o][p] = fmaf(sR * wr[p] + sI * wi[p], sE, ar[o][p]);
                                ai[o][p] = fmaf(sI * wr[p] - sR * wi[p], sE, ai[o][p]);
                            }
                        }
                        slot_i = slot_i + 1 == batchlen ? 0 : slot_i + 1;
                        u += sps;
                    }
                    if (!tap) {                                // lanes without a tap read neighbours' samples: their sums are dropped
#pragma unroll
                        for (int o = 0; o < NO; o++) ar[o][0] = ar[o][1] = ai[o][0] = ai[o][1] = 0.f;
                    }
                } else {
                int sx = sps * (kraw - batchlen + joff) + tl - mh;
                const float *x0 = x + sx, *x1 = x0 + N, *x2 = x1 + N, *x3 = x2 + N;
#pragma unroll CMA_UNROLL
                for (int kk = kraw - batchlen; kk < kraw; kk++) {
                    const float *s = ring + (size_t)slot_i * 8;
                    const bool ok = tap && sx >= 0 && sx < N;
                    const int d = ok ? 0 : -sx;                 // (masked lanes read sample 0: one unconditional load per row)
                    const float w0 = x0[d], w1 = x1[d], w2 = x2[d], w3 = x3[d];
                    const float wr[2] = {ok ? w0 * inv : 0.f, ok ? w2 * inv : 0.f}, wi[2] = {ok ? w1 * inv : 0.f, ok ? w3 * inv : 0.f};
#pragma unroll
                    for (int o = 0; o < NO; o++) {
                        const float sR = HALF ? s[ob * 2] : s[o * 2], sI = HALF ? s[ob * 2 + 1] : s[o * 2 + 1], sE = HALF ? s[4 + ob] : s[4 + o];
#pragma unroll
                        for (int p = 0; p < 2; p++) {
                            ar[o][p] = fmaf(sR * wr[p] + sI * wi[p], sE, ar[o][p]);
                            ai[o][p] = fmaf(sI * wr[p] - sR * wi[p], sE, ai[o][p]);
                        }
                    }
                    slot_i = slot_i + 1 == batchlen ? 0 : slot_i + 1;
                    sx += sps; x0 += sps; x1 += sps; x2 += sps; x3 += sps;
                }
                }
#pragma unroll
                for (int o = 0; o < NO; o++)
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        hr[o][p] = fmaf(two_lr, ar[o][p], hr[o][p]);
                        hi[o][p] = fmaf(two_lr, ai[o][p], hi[o][p]);
                    }
            }
            __syncthreads();                                   // one wave per workgroup: every read of the ring is done ...
            float *slot = ring + (size_t)(kraw % batchlen) * 8; // ... before symbol kraw replaces symbol kraw - batchlen in it
            if (lane < 4) slot[lane] = lane == 0 ? o_[0][0] : lane == 1 ? o_[0][1] : lane == 2 ? o_[1][0] : o_[1][1];   // (selects: a run-time register index would go through scratch)
            if (lane == 4) { slot[4] = e_[0]; slot[5] = e_[1]; }
            if constexpr (!STAGE) __syncthreads();
        }
        if constexpr (STAGE) {
            if (lane < sps) xs[(M + sps * (j + 1) + lane) & xmask] = nx;
            __syncthreads();
            yr[0] = yrn[0]; yr[1] = yrn[1]; yi[0] = yin[0]; yi[1] = yin[1];
        } else {
            yr[0] = yrn[0] * inv; yr[1] = yrn[1] * inv; yi[0] = yin[0] * inv; yi[1] = yin[1] * inv;
        }
    }
    if (tap) {
#pragma unroll
        for (int o = 0; o < NO; o++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                hrun[(((ob + o) * 2 + p) * 2 + 0) * M + tl] = hr[o][p];
                hrun[(((ob + o) * 2 + p) * 2 + 1) * M + tl] = hi[o][p];
            }
    }
}

constexpr int CPE_NT = 512;

__global__ __launch_bounds__(CPE_NT) void cpe_kernel(int N, int M_ma, const float *__restrict__ y, float *__restrict__ yout)
{
    extern __shared__ float cpe_lds[];                         // [3][N]: 4th power (re, im) and the phase estimate of one polarisation
    __shared__ int cnt[CPE_NT + 1];
    __shared__ float csr[CPE_NT], csi[CPE_NT];                // sums of the threads' chunks of the 4th power
    float *p4r = cpe_lds, *p4i = cpe_lds + N, *phi = cpe_lds + 2 * N;
    const int run = blockIdx.x, tid = threadIdx.x;
    // a thread owns one chunk of consecutive symbols; an ODD chunk length keeps the lanes' LDS accesses on different banks
    const int chunk = ((N + CPE_NT - 1) / CPE_NT) | 1, n0 = min(N, tid * chunk), n1 = min(N, n0 + chunk), half = M_ma / 2;
    for (int p = 0; p < 2; p++) {
        const float *a = y + ((size_t)run * 2 + p) * 2 * N, *b = a + N;
        for (int n = tid; n < N; n += CPE_NT) {               // (a + jb)^4 = a^4 - 6 a^2 b^2 + b^4 + j 4 (a^3 b - a b^3), coalesced
            const float av = a[n], bv = b[n], a2 = av * av, b2 = bv * bv;
            p4r[n] = a2 * a2 - 6.0f * a2 * b2 + b2 * b2;
            p4i[n] = 4.0f * (a2 * av * bv - av * b2 * bv);
        }
        __syncthreads();
        {                                                      // chunk sums: a window's first sum is whole chunks + two ragged ends (50 instead of 501 reads)
            float cr = 0.f, ci = 0.f;
            for (int n = n0; n < n1; n++) { cr += p4r[n]; ci += p4i[n]; }
            csr[tid] = cr; csi[tid] = ci;
        }
        __syncthreads();
        if (n0 < n1) {                                         // moving average over [n - half, n + half] (zero outside): first sum, then slide
            float sr = 0.f, si = 0.f;
            const int lo0 = max(0, n0 - half), hi0 = min(N - 1, n0 + half);
            const int c0 = (lo0 + chunk - 1) / chunk, c1 = (hi0 + 1) / chunk;      // whole chunks c0 .. c1 - 1 lie inside [lo0, hi0]
            if (c0 < c1) {
                for (int m = lo0; m < c0 * chunk; m++) { sr += p4r[m]; si += p4i[m]; }
                for (int c = c0; c < c1; c++) { sr += csr[c]; si += csi[c]; }
                for (int m = c1 * chunk; m <= hi0; m++) { sr += p4r[m]; si += p4i[m]; }
            } else {
                for (int m = lo0; m <= hi0; m++) { sr += p4r[m]; si += p4i[m]; }
            }
            for (int n = n0; n < n1; n++) {
                phi[n] = atan2f(si / (float)M_ma, -sr / (float)M_ma) * 0.25f;
                const int lo = n - half, hi_ = n + half + 1;
                if (hi_ < N) { sr += p4r[hi_]; si += p4i[hi_]; }
                if (lo >= 0) { sr -= p4r[lo]; si -= p4i[lo]; }
            }
        }
        __syncthreads();
        // unwrapping (:165-170): sample n is corrected by -pi/2 per upward jump (> pi/4) and +pi/2 per downward jump before it
        int local = 0;
        for (int n = max(n0, 1); n < n1; n++) {
            const float d = phi[n] - phi[n - 1];
            local += (d < -0.78539816339744831f) - (d > 0.78539816339744831f);
        }
        cnt[tid + 1] = local;
        __syncthreads();
        if (tid == 0) {
            cnt[0] = 0;
            for (int i = 1; i <= CPE_NT; i++) cnt[i] += cnt[i - 1];
        }
        __syncthreads();
        int acc = cnt[tid];                                    // jumps before this thread's chunk
        float prev = n0 > 0 && n0 < N ? phi[n0 - 1] : 0.f;
        for (int n = n0; n < n1; n++) {                        // phi <- unwrapped phase (only this thread touches its chunk now)
            const float ph = phi[n];
            if (n > 0) {
                const float d = ph - prev;
                acc += (d < -0.78539816339744831f) - (d > 0.78539816339744831f);
            }
            prev = ph;
            p4r[n] = ph + 1.5707963267948966f * (float)acc;    // the 4th-power buffer is spent: reuse it for the unwrapped phase
        }
        __syncthreads();
        float *oa = yout + ((size_t)run * 2 + p) * 2 * N, *ob = oa + N;
        for (int n = tid; n < N; n += CPE_NT) {               // de-rotation, coalesced
            float sn, cs;
            sincosf(p4r[n], &sn, &cs);
            const float av = a[n], bv = b[n];
            oa[n] = av * cs - bv * sn;
            ob[n] = bv * cs + av * sn;
        }
        __syncthreads();
    }
}

}  // namespace vaeq

extern "C" int vaeq_cma(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t mode, int32_t batchlen, int32_t symb_step, const float *rx,
                        float R_mod, float *h, const float *lr, float *out, float *e, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!rx || !h || !lr || !out) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || N > 0x3fffffff || sps <= 0 || M <= 0 || (M & 1) == 0 || M > 63 || N / sps < 2 * M) return VAEQ_ERR_SHAPE;
    if (!(mode == 0 || mode == 1) || (mode == 1 && (batchlen <= 0 || symb_step <= 0 || batchlen > 4096))) return VAEQ_ERR_SHAPE;
    if (mode == 0) batchlen = symb_step = 0;                   // plain CMA takes neither: whatever was passed never reaches the sizing below or the kernel
    size_t lds = mode == 1 ? (size_t)batchlen * 8 * sizeof(float) : 0;
    int xcap = 64;                                             // sample ring: a power of two >= sps (batchlen + 2) + M positions
    while (mode == 1 && xcap < sps * (batchlen + 2) + M) xcap <<= 1;   // (mode 1: batchlen <= 4096 was checked above, so this ends)
    // staged samples pay when the 100-term update runs often (CMAflex: every symb_step = 10 symbols); with one update per batchlen symbols (CMAbatch)
    // the per-symbol barrier of the staged form costs more than the update gains (17.7 vs 20.3 ms per 8192-run frame)
    const bool stage = mode == 1 && lds + (size_t)xcap * 16 <= 24 * 1024 && 4 * symb_step <= batchlen;
    if (stage) lds += (size_t)xcap * 16;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    void (*k)(int, int, int, int, int, int, int, const float *, float, float *, const float *, float *, float *) =
        M <= 32 ? (stage ? vaeq::cma_kernel<true, true> : vaeq::cma_kernel<true, false>) : (stage ? vaeq::cma_kernel<false, true> : vaeq::cma_kernel<false, false>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    hipLaunchKernelGGL(k, dim3(R), dim3(64), lds, st, (int)N, sps, M, mode, batchlen, symb_step, xcap - 1, rx, R_mod, h, lr, out, e);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_cpe(int32_t R, int64_t N, int32_t M_ma, const float *y, float *y_out, void *stream)
{
    if (R == 0 || N == 0) return VAEQ_OK;
    if (!y || !y_out) return VAEQ_ERR_NULL;
    if (R < 0 || N < 0 || N > 12800 || M_ma <= 0 || (M_ma & 1) == 0) return VAEQ_ERR_SHAPE;   // three N-float tracks of one polarisation live in LDS
    const size_t lds = (size_t)3 * N * sizeof(float);
    auto k = vaeq::cpe_kernel;
    if (lds > 32 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    hipLaunchKernelGGL(k, dim3(R), dim3(vaeq::CPE_NT), lds, reinterpret_cast<hipStream_t>(stream), (int)N, M_ma, y, y_out);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}
