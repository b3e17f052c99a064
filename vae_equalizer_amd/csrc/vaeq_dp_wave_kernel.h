// vaeq_dp_wave_kernel.h -- wave-per-run fast path of the DP VAE-LE / VAEflex training loop (gfx950).
//
// One 64-lane wavefront = one run (B <= 128); no workgroup barriers, everything between HBM and the taps lives in LDS/VGPRs.
// Same math as vaeq_dp.hip (see the derivation there and in DESIGN.md); what changes is the mapping:
//
//   * lane l owns the symbol pair (2l, 2l+1) of the minibatch (B <= 128): FIR, soft demap, the per-symbol moments
//     (kept in registers until the backward pass) and dL/dy are all done by the owning lane; q and y leave as one
//     8-byte store per lane and row (400 contiguous bytes per row and step at B = 100).
//   * every convolution-shaped phase is register-blocked over that pair and over both outputs: the taps of a group of four are read once
//     (LDS broadcasts) and feed 32 packed FMAs
//     (FIR:  y[n]      = sum_k w[k] x[2n+k],
//      dL/dU[n]        = sum_j e[2n+j] conj(h[j])    -- the same shape on the residual e, both chi in one loop,
//      D[t], t=4l..4l+3: the zero-stuffed convolution, written as two polyphase symbol-rate FIRs on mu).
//   * sample-rate arrays (x, e) are stored 4-way polyphase in LDS (index c -> [c & 3][c >> 2]) and the symbol-rate mu
//     2-way, so that the stride-4 / stride-2 accesses of consecutive lanes hit consecutive 8-byte LDS words.
//   * the two correlation-shaped gradients (dL/dh: 100 outputs x 88 terms, dL/dw: 100 x 100) are laid out as
//     lane = (tap, half of the sum range); halves are combined with one cross-half shuffle; the lane that ends up
//     with a gradient also owns that parameter's Adam moments (LDS rows of their own) and writes the updated tap to LDS.
//   * every 8-byte LDS read is pinned to one ds_read_b64 (lds2), the tap loops are two-deep software pipelines (pipe2) whose first
//     term starts the accumulators; streamed rows go through buffer descriptors (scalar row offsets, out-of-range offsets as masks).
//   * wave sums and prefix scans run on DPP in a fixed order: bitwise reproducible, no LDS round trips.
//   (DESIGN.md section 5 has the measurements behind each of these.)
//
// Supported here: sps == 2, B even, 2*(M/2)+2 <= B <= 1024, M in {9, 13, 17, 21, 25, 31}; everything else takes the
// generic kernel of vaeq_dp.hip.  BT > 0 bakes the minibatch length into the kernel (all LDS offsets immediate).
// B <= 128 runs as ONE wavefront per run (NW = 1, described above); 128 < B <= 256 as two, B <= 512 as four and B <= 1024 as
// eight wavefronts per run (NW = 2, 4, 8: same code, thread 64 wv + lane owns the pair, see dp_wave_kernel).  Instantiated in
// vaeq_dp_wave.hip (NW = 1), vaeq_dp_wave_mw.hip (NW = 2, 4) and vaeq_dp_wave_mw8.hip (NW = 8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "vaeq.h"
#include "vaeq_common.h"
#include "vaeq_wave.h"

// -DVAEQ_PHASE_STAMPS: shader-clock stamps around the phases of the LAST step of one wave under full load, written over loss[0..] of the launch
// (tools/probe_dp_phases.py reads them; the build is for that probe only: tools/snap_variant.sh stamps -DVAEQ_PHASE_STAMPS).
#ifdef VAEQ_PHASE_STAMPS
#define VAEQ_STAMP(i) do { if (s == a.steps - 1) tst[i] = __builtin_readcyclecounter(); } while (0)
#define VAEQ_XSTAMP(i) do { if (s == a.steps - 1) tsx[i] = __builtin_readcyclecounter(); } while (0)   // inside the two tap-gradient phases
#else
#define VAEQ_STAMP(i) do { } while (0)
#define VAEQ_XSTAMP(i) do { } while (0)
#endif
#define VAEQ_NSTAMP 12
// Experiment knobs of tools/snap_variant.sh builds (the defaults are what ships): software pipelining of the tap loops / of dL/dU
#ifndef VAEQ_PIPE
#define VAEQ_PIPE 1
#endif
#ifndef VAEQ_PIPE_DU
#define VAEQ_PIPE_DU 0
#endif
#ifndef VAEQ_ROW_STEP
#define VAEQ_ROW_STEP 1                                // q row offsets as one running scalar (0: products row * No4, A/B knob)
#endif
#define VAEQ_BT_FRAMES 16                              // frames per launch whose Adam bias corrections restart exactly like a launch's (vaeq_dp_train)
#ifndef VAEQ_DEMAP_SHIFT
#define VAEQ_DEMAP_SHIFT 0                             // 1: softmax shift from the nearest level instead of a maximum search -- 32 instructions fewer per step, but
                                                       // the fused form spills 7 VGPRs at the 254-register limit: 8.65 vs 8.45 ms (profiles/r03/kernel_variants_ab.txt); off
#endif
#ifndef VAEQ_WPS
#define VAEQ_WPS 2                                     // workgroups per SIMD the register budget is sized for (3 would need <= 168 VGPRs: it spills)
#endif

namespace vaeq {

struct WaveLayout {
    int Lph, Uph;                                      // float2 per polyphase component of x/e and of mu
    int X, E, U, PSv, W, H, PSh, VS, MOM, RED, XG, total;   // byte offsets (the dL/dy buffer aliases U)
};

__host__ __device__ inline WaveLayout wave_layout(int B, int M, int NW = 1)
{
    WaveLayout l;
    const int len = 2 * B + M - 1;                     // L + 2*mh samples incl. zero halo
    const int lph = wave_lph(len);
    l.Lph = lph;
    l.Uph = B / 2 + 1;
    int o = 0;
    auto take = [&](int bytes) { int r = o; o += (bytes + 15) & ~15; return r; };
    l.X = take(2 * 4 * lph * 8);
    l.E = take(2 * 4 * lph * 8);
    const int ubytes = 2 * 2 * l.Uph * 8, gbytes = 2 * B * 8;
    l.U = take(ubytes > gbytes ? ubytes : gbytes);
    l.PSv = take(2 * (B + 1) * 4);
    l.W = take(2 * M * 16);
    l.H = take(2 * 2 * (M + 1) * 8);                   // one zero pad tap per (chi, nu): j = M
    l.PSh = take(2 * (M + 1) * 4);
    l.VS = take(2 * M * 4);
    l.MOM = take(16 * 64 * 4);                         // Adam moments of the taps a lane owns: [16][64 lanes] (registers are the scarcer resource)
    l.RED = take(NW > 1 ? 64 * 4 : 0);                 // NW > 1: cross-wave scan offsets and sums
    l.XG = take((NW - 1) * 4 * 64 * 4);                // ... and the tap-gradient partial sums of waves 1..NW-1
    l.total = o;
    return l;
}

// y[sym] += sum_k taps[k] * x[4l + 2*sym + k] for the lane's symbol pair, one input polarisation (FIR): T = float4 tap quads
// (o0.re, o0.im, o1.re, o1.im), acc[sym][o] += w * x.  xp = phase-0 pointer of the lane (slot = lane).
// A dependent v_pk_fma_f32 can issue only every ~12 cycles (measured: one wave with 8 accumulator chains reaches 64 % of the packed-FMA rate,
// tools/ubench/coissue.hip), and with 2 waves per SIMD a wave is often alone in its FMA burst -- so every convolution-shaped phase keeps 16
// independent chains: here the even taps accumulate into acc, the odd taps into acc2 (the caller adds them once at the end).
// INIT: the first taps start the accumulators (no zeroing); else they are added to.
// SPLIT = false (shapes not baked into the kernel, where run-time strides already cost registers): all taps into acc, acc2 untouched.
template <int M, bool INIT, bool PIPE = true, bool SPLIT = true>
__device__ __forceinline__ void pair_fir(cacc (&acc)[2][2], cacc (&accx)[2][2], const float2 *xp, int Lph, const float4 *wq)
{
    cacc (&acc2)[2][2] = SPLIT ? accx : acc;
    auto tapA = [&](int k) -> v2f { return lds2(reinterpret_cast<const float2 *>(wq + k)); };       // (re, im) of o = 0
    auto tapB = [&](int k) -> v2f { return lds2(reinterpret_cast<const float2 *>(wq + k) + 1); };   // (re, im) of o = 1
    constexpr int G = M / 4;
    // stage g: taps 4g..4g+3, samples c' = 4g..4g+5 -- 14 independent 8-byte LDS reads feeding 32 packed FMAs
    auto load = [&](int g, v2f (&r)[14]) {
        const float2 *xg = xp + g;
        r[0] = lds2(xg); r[1] = lds2(xg + Lph); r[2] = lds2(xg + 2 * Lph); r[3] = lds2(xg + 3 * Lph); r[4] = lds2(xg + 1); r[5] = lds2(xg + Lph + 1);
#pragma unroll
        for (int t = 0; t < 4; t++) { r[6 + 2 * t] = tapA(4 * g + t); r[7 + 2 * t] = tapB(4 * g + t); }
    };
    auto fma = [&](int, const v2f (&r)[14], auto first) {
        constexpr bool F = decltype(first)::value;     // the very first taps start the accumulators (cmul): nothing to zero
        cmacf<F>(acc[0][0], r[6], r[0]); cmacf<F>(acc[0][1], r[7], r[0]); cmacf<F>(acc[1][0], r[6], r[2]); cmacf<F>(acc[1][1], r[7], r[2]);
        cmacf<F && SPLIT>(acc2[0][0], r[8], r[1]); cmacf<F && SPLIT>(acc2[0][1], r[9], r[1]);
        cmacf<F && SPLIT>(acc2[1][0], r[8], r[3]); cmacf<F && SPLIT>(acc2[1][1], r[9], r[3]);
        cmac(acc[0][0], r[10], r[2]); cmac(acc[0][1], r[11], r[2]); cmac(acc[1][0], r[10], r[4]); cmac(acc[1][1], r[11], r[4]);
        cmac(acc2[0][0], r[12], r[3]); cmac(acc2[0][1], r[13], r[3]); cmac(acc2[1][0], r[12], r[5]); cmac(acc2[1][1], r[13], r[5]);
    };
    pipe2<14, INIT, PIPE>(G, load, fma);
#pragma unroll
    for (int k = 4 * G; k < M; k++) {                  // remaining 1 or 3 taps
        const v2f a = tapA(k), b = tapB(k);
        const v2f xa = lds2(xp + (k & 3) * Lph + (k >> 2)), xb = lds2(xp + ((k + 2) & 3) * Lph + ((k + 2) >> 2));
        if (k & 1) { cmac(acc2[0][0], a, xa); cmac(acc2[0][1], b, xa); cmac(acc2[1][0], a, xb); cmac(acc2[1][1], b, xb); }
        else { cmac(acc[0][0], a, xa); cmac(acc[0][1], b, xa); cmac(acc[1][0], a, xb); cmac(acc[1][1], b, xb); }
    }
}

// dL/dU for the lane's symbol pair: the FIR's shape on the residual e with conjugated channel taps, acc[chi][sym][nu] += e[chi] * conj(h[chi][nu]),
// BOTH chi in one loop (16 independent accumulator chains, see pair_fir).  ep = phase-0 pointer of the lane into e[chi = 0] (chi = 1: + 4 Lph);
// ht[chi * 2 + nu] = the tap rows (float2, MP apart).  Runs at the kernel's register peak (the demapper's moments are live): one operand set.
template <int M, bool PIPE = false, bool MERGE = true>
__device__ __forceinline__ void pair_du(cacc (&acc)[2][2][2], const float2 *ep, int Lph, const float2 *ht, int MP)
{
    constexpr int G = M / 4;
    // operands of one chi for taps 4g..4g+3: 6 samples c' = 4g..4g+5 and the 4 taps of the rows nu = 0, 1 -- 14 independent LDS reads, 32 packed FMAs
    auto load1 = [&](int chi, int g, v2f *x) {
        const float2 *xg = ep + chi * 4 * Lph + g;
        x[0] = lds2(xg); x[1] = lds2(xg + Lph); x[2] = lds2(xg + 2 * Lph); x[3] = lds2(xg + 3 * Lph); x[4] = lds2(xg + 1); x[5] = lds2(xg + Lph + 1);
#pragma unroll
        for (int t = 0; t < 4; t++) { x[6 + 2 * t] = lds2(ht + (chi * 2 + 0) * MP + 4 * g + t); x[7 + 2 * t] = lds2(ht + (chi * 2 + 1) * MP + 4 * g + t); }
    };
    auto fma1 = [&](int chi, const v2f *x, int t, auto first) {                // tap 4g + t of both rows on both symbols
        constexpr bool F = decltype(first)::value;
        cmacf<F>(acc[chi][0][0], x[6 + 2 * t], x[t]); cmacf<F>(acc[chi][0][1], x[7 + 2 * t], x[t]);
        cmacf<F>(acc[chi][1][0], x[6 + 2 * t], x[t + 2]); cmacf<F>(acc[chi][1][1], x[7 + 2 * t], x[t + 2]);
    };
    if constexpr (MERGE) {                             // both chi per stage: 28 reads feeding 64 FMAs on 16 chains
        auto load = [&](int g, v2f (&r)[28]) { load1(0, g, r); load1(1, g, r + 14); };
        auto fma = [&](int, const v2f (&r)[28], auto first) {
            fma1(0, r, 0, first); fma1(1, r + 14, 0, first);
#pragma unroll
            for (int t = 1; t < 4; t++) { fma1(0, r, t, std::false_type{}); fma1(1, r + 14, t, std::false_type{}); }
        };
        pipe2<28, true, PIPE>(G, load, fma);
    } else {                                           // one chi after the other (half the operand registers)
#pragma unroll
        for (int chi = 0; chi < 2; chi++) {
            auto load = [&](int g, v2f (&r)[14]) { load1(chi, g, r); };
            auto fma = [&](int, const v2f (&r)[14], auto first) {
                fma1(chi, r, 0, first);
#pragma unroll
                for (int t = 1; t < 4; t++) fma1(chi, r, t, std::false_type{});
            };
            pipe2<14, true, PIPE>(G, load, fma);
        }
    }
#pragma unroll
    for (int k = 4 * G; k < M; k++)                    // remaining 1 or 3 taps
#pragma unroll
        for (int chi = 0; chi < 2; chi++) {
            const float2 *xp = ep + chi * 4 * Lph;
            const v2f a = lds2(ht + (chi * 2 + 0) * MP + k), b = lds2(ht + (chi * 2 + 1) * MP + k);
            const v2f xa = lds2(xp + (k & 3) * Lph + (k >> 2)), xb = lds2(xp + ((k + 2) & 3) * Lph + ((k + 2) >> 2));
            cmac(acc[chi][0][0], a, xa); cmac(acc[chi][0][1], b, xa); cmac(acc[chi][1][0], a, xb); cmac(acc[chi][1][1], b, xb);
        }
}

// Whether the dL/dh (T = B - mh terms per parity) and dL/dw (B / 2 pairs) sums split into NP parts of equal, non-zero length.
constexpr bool uniform_parts(int B, int M, int NP)
{
    const int mh = M / 2, nm = 2 * B - 2 * mh;
    for (int par = 0; par < 2; par++) {
        const int T = (nm - par + 1) >> 1, Th = ((T + 2 * NP - 1) / (2 * NP)) << 1;
        int n0 = -1;
        for (int part = 0; part < NP; part++) {
            const int ma = (part * Th) >> 1, e = part * Th + Th, mb = ((T < e ? T : e) + 1) >> 1;
            if (mb - ma <= 0 || (n0 >= 0 && mb - ma != n0)) return false;
            n0 = mb - ma;
        }
    }
    const int Bq = ((B + 2 * NP - 1) / (2 * NP)) << 1;
    int n0 = -1;
    for (int part = 0; part < NP; part++) {
        const int ma = (part * Bq) >> 1, e = part * Bq + Bq, mb = (B < e ? B : e) >> 1;
        if (mb - ma <= 0 || (n0 >= 0 && mb - ma != n0)) return false;
        n0 = mb - ma;
    }
    return true;
}

// OUT: 0 = every output nullable at run time; 1 = no compact outputs (eq_out / dec_out ignored); 2 = compact outputs, q not written.
// The specialisations only drop dead code: fewer live scalars, fewer SGPR spills in the step loop.
// NW = wavefronts per run: 1 (B <= 128, no barriers at all) or 2 / 4 / 8 (B <= 256 / 512 / 1024): thread gl = 64 wv + lane owns the symbol pair
// (2 gl, 2 gl + 1), the tap-gradient sums are split 2 NW ways, wave 0 owns the taps and their Adam moments; phases are separated
// by s_barrier after an LDS-only wait (sync_lds), so the in-flight q / y stores still never stall a phase.
template <int M, int NLEV, int BT, bool PAIR, int OUT, int NW = 1, int BL = 0>
__global__ __launch_bounds__(64 * NW, VAEQ_WPS) void dp_wave_kernel(const vaeq_dp_args a)
{
    constexpr int mh = M / 2, Mh = 2 * mh, MP = M + 1;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    extern __shared__ float4 smem4[];
    char *sm = reinterpret_cast<char *>(smem4);
    const int gl = threadIdx.x, lane = NW > 1 ? (gl & 63) : gl, wv = NW > 1 ? (gl >> 6) : 0, run = blockIdx.x;
    constexpr int NT = 64 * NW, NP = 2 * NW;              // threads per run; parts a tap-gradient sum is split into
    // baked shape whose tap-gradient sums split into NP equal, non-empty parts: their loops run on a scalar trip count and start from the first term
    constexpr bool UNI = BT > 0 && uniform_parts(BT, M, NP);
    // BL > 0 (with BT == 0): the LDS LAYOUT of minibatch length BL (all offsets and strides immediate) for any run-time B <= BL of BL's parity class
    // (M = 13 / 17 on a fixed layout: the pipelined form unrolls into 27-117 spilled registers and loses 25-38 %: they keep the lean loops; so
    //  does M = 31 at one wavefront per run, where it costs 2-5 %)
    constexpr bool FIXL = BT > 0 || (BL > 0 && M != 13 && M != 17 && !(M == 31 && NW == 1));
    constexpr bool PIPE = VAEQ_PIPE && FIXL;               // run-time layouts keep more addresses live: there one operand set,
    constexpr bool WIDE = FIXL;                            // ... 8 accumulator chains and one chi at a time in dL/dU (fits the register file)
    constexpr bool PIPE_DU = VAEQ_PIPE_DU;                 // dL/dU runs at the kernel's register peak (moments of the demapper still live): no second operand set there
    const int B = BT ? BT : a.B;
    const int BS = BT ? BT : BL ? BL : B;                      // the minibatch length the LDS layout (offsets, row strides) is made for
    // ODDB: on a class layout (BL) the minibatch length may be odd -- the last lane's pair is (B - 1, phantom): the phantom symbol is computed like
    // any other (from zero-padded cells) and masked wherever it would be stored or summed (v1 below); baked shapes are even and unchanged
    constexpr bool ODDB = BT == 0 && BL > 0;
    const int L = 2 * B, nm = L - Mh, P2 = ODDB ? (B + 1) / 2 : B / 2;
    const float rnm = 1.0f / (float)nm;
    const WaveLayout lay = wave_layout(BS, M, NW);
    const int Lph = lay.Lph, Uph = lay.Uph;
    float2 *Xs = reinterpret_cast<float2 *>(sm + lay.X), *Es = reinterpret_cast<float2 *>(sm + lay.E);
    float2 *Us = reinterpret_cast<float2 *>(sm + lay.U), *GY = Us;
    float4 *Wt = reinterpret_cast<float4 *>(sm + lay.W);       // [p][k] = (wr[o0], wi[o0], wr[o1], wi[o1])
    float2 *Ht = reinterpret_cast<float2 *>(sm + lay.H);       // [chi][nu][j] = (re, im), j = 0..M (j = M: zero pad)
    float *PSv = reinterpret_cast<float *>(sm + lay.PSv);      // [nu][B+1] exclusive prefix sums of v_I + v_Q
    float *PSh = reinterpret_cast<float *>(sm + lay.PSh);      // [nu][M+1] exclusive prefix sums of sum_chi gC |h|^2
    float *VS = reinterpret_cast<float *>(sm + lay.VS);        // [nu][M]
    float *RED = reinterpret_cast<float *>(sm + lay.RED), *XG = reinterpret_cast<float *>(sm + lay.XG);   // NW > 1 only

    // ---- per-run constants (uniform)
    float amp[NLEV], b2[NLEV], nlogP[NLEV];
    const float nusc = a.nu_sc[run];
#pragma unroll
    for (int i = 0; i < NLEV; i++) {
        amp[i] = a.amp[i];
        b2[i] = nusc * amp[i] * amp[i] * LOG2E;
        nlogP[i] = -logf(a.P[(size_t)run * NLEV + i]);
    }
    const float var0 = a.var[run * 2 + 0], var1 = a.var[run * 2 + 1];
    const float lrW = a.lr_W[run], lrH = a.lr_h[run];
    const float lev_delta = amp[NLEV - 1] - amp[NLEV - 2], lev_inv = 1.0f / lev_delta, lev_off = -amp[0] * lev_inv;   // the (equidistant) level grid

    // ---- zero the halo'd buffers once; load taps; owner lanes load their Adam moments
    for (int i = gl; i < (lay.W - lay.X) / 8; i += NT) Xs[i] = make_float2(0.f, 0.f);     // X, E, U, PSv
    for (int i = gl; i < (lay.PSh - lay.H) / 8; i += NT) Ht[i] = make_float2(0.f, 0.f);   // incl. the pad taps
    __syncthreads();
    const int tk = lane & 31, half = lane >> 5;                // tap index / which half this lane owns (o resp. chi)
    const bool worker = tk < M, owner = worker && wv == 0;     // worker: sums a part of a tap's gradient; owner: holds the tap
    const int part = wv * 2 + half;
    const size_t gbase = (size_t)run * 8 * M;
    // Adam moments of the taps this lane owns live in LDS, one 64-lane row per quantity (touched twice per step; 16 VGPRs freed):
    // rows 0-7: m, v of W[o=half][p][k=tk] (re, im); rows 8-15: m, v of h[chi=half][nu][.][j=tk]
    float *MOM = reinterpret_cast<float *>(sm + lay.MOM) + lane;
#define mWr(p) MOM[(0 + (p)) * 64]
#define mWi(p) MOM[(2 + (p)) * 64]
#define vWr(p) MOM[(4 + (p)) * 64]
#define vWi(p) MOM[(6 + (p)) * 64]
#define mHr(p) MOM[(8 + (p)) * 64]
#define mHi(p) MOM[(10 + (p)) * 64]
#define vHr(p) MOM[(12 + (p)) * 64]
#define vHi(p) MOM[(14 + (p)) * 64]
    if (owner) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const size_t ir = gbase + (half * 4 + p) * M + tk, ii = gbase + (half * 4 + 2 + p) * M + tk;
            float *wq = reinterpret_cast<float *>(&Wt[p * M + tk]);
            wq[half * 2 + 0] = a.W[ir];
            wq[half * 2 + 1] = a.W[ii];
            mWr(p) = a.adam_mW[ir]; mWi(p) = a.adam_mW[ii];
            vWr(p) = a.adam_vW[ir]; vWi(p) = a.adam_vW[ii];
            const size_t hr = gbase + ((half * 2 + p) * 2 + 0) * M + tk, hi = hr + M;
            Ht[(half * 2 + p) * MP + tk] = make_float2(a.h[hr], a.h[hi]);
            mHr(p) = a.adam_mh[hr]; mHi(p) = a.adam_mh[hi];
            vHr(p) = a.adam_vh[hr]; vHi(p) = a.adam_vh[hi];
        }
    }
    int step = a.step[run];
    // beta^t at the start of each of the launch's first VAEQ_BT_FRAMES frames, evaluated here, where no register is under pressure yet (one pow call
    // site, a lane per frame), and picked up at the frame heads; inside a frame it is a running product in double
    __shared__ double BTW[2][VAEQ_BT_FRAMES];
    if (gl < VAEQ_BT_FRAMES && gl < a.n_frames) {
        const double t = (double)(step + gl * a.steps);
        BTW[0][gl] = pow(0.9, t);
        BTW[1][gl] = pow(0.999, t);
    }
    __syncthreads();
    double b1t = BTW[0][0], b2t = BTW[1][0];

    const int klen = a.keep_len, k0 = a.keep_off;
    const size_t No = (size_t)a.steps * klen;
    constexpr bool pairst = PAIR;                              // keep_off, keep_len even: both symbols of a lane kept together -> 8-byte stores
    const bool act = gl < P2;                                  // thread owns symbols 2*gl, 2*gl+1
    const int nq = (nm + 3) / 4;                               // residual quads t = 4l' .. 4l'+3
    const int n0 = 2 * gl;
    const bool v1 = ODDB ? act && (n0 + 1 < B) : act;       // the lane's second symbol exists
    const float2 *Xl = Xs + gl, *El = Es + gl, *Ul = Us + gl;
    const float2 *Xa = Xs + (act ? gl : 0), *Ea = Es + (act ? gl : 0);   // symbol-pair phases: idle lanes shadow lane 0 (results unused)

    // The window of the NEXT step is fetched into registers while the current step computes (one 16-byte load per lane and
    // row: B <= 128 means L/4 <= 64 lanes), so a step never waits for HBM after the first.
    const bool ldl = gl < (ODDB ? (L + 3) / 4 : L / 4);         // odd B: L = 2 (mod 4), the last lane's upper two samples belong to the next minibatch
    float4 pf[4];
    const uint32_t S4 = (uint32_t)a.S * 4u;                    // bytes per received row; a frame of a run = 4 rows
    auto frame_rsrc = [&](int f) { return make_rsrc(a.rx + ((size_t)run * a.n_frames + f) * 4 * (size_t)a.S, 4u * S4); };
    auto fetch = [&](const __amdgpu_buffer_rsrc_t xr, int s) {
        const uint32_t vo = ldl ? ((uint32_t)s * (uint32_t)a.stride_sym * 2u + 4u * gl) * 4u : OOB;   // lanes beyond the window read zeros
#pragma unroll
        for (int r = 0; r < 4; r++) pf[r] = bld128(xr, vo, (uint32_t)r * S4);
    };
#ifdef VAEQ_PHASE_STAMPS
    long long tst[VAEQ_NSTAMP], tsx[8];
#endif
    fetch(frame_rsrc(0), 0);
    for (int f = 0; f < a.n_frames; f++) {
        const __amdgpu_buffer_rsrc_t xr = frame_rsrc(f), xn = frame_rsrc(f + 1 < a.n_frames ? f + 1 : f);   // this frame's rows, the next frame's
        if (f && f < VAEQ_BT_FRAMES) {                         // beta^t restarts from pow() at every frame, as it does at a launch: how many frames a
            b1t = BTW[0][f];                                    // launch holds is a scheduling choice, the results are bit-identical either way
            b2t = BTW[1][f];                                    // (dp_runs.run_dp_batch groups the frames of small sweeps)
        }
        // one buffer descriptor per output array and frame; rows are addressed by scalar offsets (row * No4) folded into the stores
        const bool qf = OUT != 2 && a.q_out, yf = a.y_out, ef = OUT != 1 && a.eq_out, df = OUT != 1 && a.dec_out;
        const uint32_t No4 = (uint32_t)No * 4u;
        const size_t fr = (size_t)run * a.n_frames + f;
        const __amdgpu_buffer_rsrc_t qr = make_rsrc(qf ? a.q_out + fr * (4 * NLEV) * No : nullptr, qf ? 4u * NLEV * No4 : 0u);
        const __amdgpu_buffer_rsrc_t yr = make_rsrc(yf ? a.y_out + fr * 4 * No : nullptr, yf ? 4u * No4 : 0u);
        const __amdgpu_buffer_rsrc_t er = make_rsrc(ef ? a.eq_out + fr * 2 * No : nullptr, ef ? 2u * No4 : 0u);
        const __amdgpu_buffer_rsrc_t dr = make_rsrc(df ? a.dec_out + fr * 4 * No : nullptr, df ? (uint32_t)(4 * No) : 0u);
        const __amdgpu_buffer_rsrc_t lr_ = make_rsrc(a.loss ? a.loss + fr * a.steps : nullptr, a.loss ? (uint32_t)a.steps * 4u : 0u);
        const __amdgpu_buffer_rsrc_t vr_ = make_rsrc(a.var_est ? a.var_est + fr * 2 * a.steps : nullptr, a.var_est ? (uint32_t)a.steps * 8u : 0u);
#pragma unroll 1
        for (int s = 0; s < a.steps; s++) {
            VAEQ_STAMP(0);
            // ============ P0: prefetched window -> LDS (polyphase scatter; halo stays zero)
            if (ldl) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const float4 I4 = pf[p * 2 + 0], Q4 = pf[p * 2 + 1];
                    const float xi[4] = {I4.x, I4.y, I4.z, I4.w}, xq[4] = {Q4.x, Q4.y, Q4.z, Q4.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int c = mh + i;                  // + 4*lane: phase (c & 3) is lane independent
                        const bool in = !ODDB || 4 * gl + i < L;
                        Xs[(p * 4 + (c & 3)) * Lph + gl + (c >> 2)] = in ? make_float2(xi[i], xq[i]) : make_float2(0.f, 0.f);
                    }
                }
            }
            sync_lds<NW>();

            VAEQ_STAMP(1);
            // ============ P1: FIR for the lane's symbol pair, both output polarisations
            float2 y[2][2];                                    // [sym][o]
            {
                cacc ya[2][2], yb[2][2];                       // lanes without a symbol pair recompute lane 0's (Xa): no divergence, nothing to zero
                pair_fir<M, true, PIPE, WIDE>(ya, yb, Xa, Lph, Wt);
                pair_fir<M, false, PIPE, WIDE>(ya, yb, Xa + 4 * Lph, Lph, Wt + M);
#pragma unroll
                for (int sy = 0; sy < 2; sy++)
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        if constexpr (WIDE) {
                            ya[sy][o].a += yb[sy][o].a;
                            ya[sy][o].b += yb[sy][o].b;
                        }
                        y[sy][o] = cfin(ya[sy][o]);
                    }
            }
            // pin: hipcc otherwise sinks whole FMA chains to their (much later) use and keeps their inputs alive instead
            asm volatile("" : "+v"(y[0][0].x), "+v"(y[0][0].y), "+v"(y[0][1].x), "+v"(y[0][1].y), "+v"(y[1][0].x), "+v"(y[1][0].y),
                         "+v"(y[1][1].x), "+v"(y[1][1].y));
            VAEQ_STAMP(2);
            const bool kept0 = act && (n0 >= k0) && (n0 < k0 + klen), kept1 = act && (n0 + 1 >= k0) && (n0 + 1 < k0 + klen);
            const uint32_t col = (uint32_t)(s * klen + (n0 - k0));
            const uint32_t vo0 = kept0 ? col * 4u : OOB, vo1 = kept1 ? col * 4u + 4u : OOB;     // byte offsets inside a row; OOB = not stored
            if (yf) {
                uint32_t yrow = 0u;                            // rows (o, re / im) in ascending order: a running scalar offset (see qrow below)
                auto next_row = [&]() {
#if VAEQ_ROW_STEP
                    yrow += No4;
                    asm volatile("" : "+s"(yrow));
#endif
                };
#pragma unroll
                for (int o = 0; o < 2; o++) {
#if VAEQ_ROW_STEP
                    const uint32_t r0 = yrow;
                    next_row();
                    const uint32_t r1 = yrow;
                    next_row();
#else
                    const uint32_t r0 = (uint32_t)(o * 2 + 0) * No4, r1 = (uint32_t)(o * 2 + 1) * No4;
#endif
                    if (pairst) {
                        bst64(v2f{y[0][o].x, y[1][o].x}, yr, vo0, r0);
                        bst64(v2f{y[0][o].y, y[1][o].y}, yr, vo0, r1);
                    } else {
                        bst32(y[0][o].x, yr, vo0, r0); bst32(y[0][o].y, yr, vo0, r1);
                        bst32(y[1][o].x, yr, vo1, r0); bst32(y[1][o].y, yr, vo1, r1);
                    }
                }
            }

            VAEQ_STAMP(3);
            // ============ P2: soft demap + moments (registers), mu -> LDS, prefix sums of the variances
            float mv[2][2][2], mt3[2][2][2], mkc[2][2][2];     // [sym][o][c]: Var_q, 3rd central moment, KL-gradient moment
#if VAEQ_ROW_STEP
            // the 4 n rows of q leave in ascending order: their byte offset is ONE running scalar (an s_add per row) -- as 32 products row * No4 they
            // were 32 live scalars, spilled and fetched back with v_readlane + hazard s_nops in front of every store
            uint32_t qrow = 0u;
#endif
            float klsum = 0.f, vv[2][2];                       // vv[o][sym] = v_I + v_Q
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const float c2 = 0.5f / (o ? var1 : var0) * LOG2E;
                float2 muv[2];                                 // per sym: (mu_I, mu_Q)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    // both symbols of the lane at once: every add/mul/fma below is one packed instruction (v_pk_*_f32)
                    const v2f yy = c ? v2f{y[0][o].y, y[1][o].y} : v2f{y[0][o].x, y[1][o].x};
                    v2f z[NLEV], q[NLEV];
                    v2f ssum = {0.f, 0.f};
#if VAEQ_DEMAP_SHIFT
                    // softmax shift without a maximum search: the squared distance to the NEAREST level (found by rounding on the equidistant level
                    // grid) is taken out of every d^2 -- z_i = -c2 (d_i^2 - dmin^2) - b2_i <= 0 for every level and >= -max b2 for the nearest one
                    // (no overflow, the sum never underflows); it rides in the FMA that squares d: 8 subtractions and 8 maxima per axis gone
                    v2f dm2;
                    {
                        const v2f t = yy * lev_inv + lev_off;  // level index coordinate (y - amp[0]) / delta
                        const v2f r = {__builtin_amdgcn_fmed3f(__builtin_rintf(t.x), 0.f, (float)(NLEV - 1)), __builtin_amdgcn_fmed3f(__builtin_rintf(t.y), 0.f, (float)(NLEV - 1))};
                        const v2f dm = (t - r) * lev_delta;
                        dm2 = dm * dm;
                    }
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        const v2f d = yy - amp[i];
                        z[i] = -((d * d - dm2) * c2 + b2[i]);
                        q[i] = v2f{__builtin_amdgcn_exp2f(z[i].x), __builtin_amdgcn_exp2f(z[i].y)};
                        ssum += q[i];
                    }
#else
                    float zm0 = -3.0e38f, zm1 = -3.0e38f;
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        const v2f d = yy - amp[i];
                        z[i] = -(d * d * c2 + b2[i]);
                        zm0 = fmaxf(zm0, z[i].x);
                        zm1 = fmaxf(zm1, z[i].y);
                    }
                    const v2f zmax = {zm0, zm1};
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        z[i] -= zmax;
                        q[i] = v2f{__builtin_amdgcn_exp2f(z[i].x), __builtin_amdgcn_exp2f(z[i].y)};
                        ssum += q[i];
                    }
#endif
                    const v2f rs = {__builtin_amdgcn_rcpf(ssum.x), __builtin_amdgcn_rcpf(ssum.y)};
                    v2f m1 = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        q[i] *= rs;
                        m1 += q[i] * amp[i];
                    }
                    // compact stand-ins for q in the epilogue: E_q[x_I] and the first maximum of q, exactly as it would derive them
                    if (ef && c == 0) {
                        if (pairst) bst64(m1, er, vo0, (uint32_t)o * No4);
                        else { bst32(m1.x, er, vo0, (uint32_t)o * No4); bst32(m1.y, er, vo1, (uint32_t)o * No4); }
                    }
                    if (df) {
                        float v0 = q[0].x, v1 = q[0].y;
                        int b0 = 0, b1 = 0;
#pragma unroll
                        for (int i = 1; i < NLEV; i++) {
                            if (q[i].x > v0) { v0 = q[i].x; b0 = i; }
                            if (q[i].y > v1) { v1 = q[i].y; b1 = i; }
                        }
                        const uint32_t ro = (uint32_t)(o * 2 + c) * (uint32_t)No;                 // one byte per decision
                        if (pairst) bst16((unsigned short)(b0 | (b1 << 8)), dr, kept0 ? col : OOB, ro);
                        else { bst8((unsigned char)b0, dr, kept0 ? col : OOB, ro); bst8((unsigned char)b1, dr, kept1 ? col + 1u : OOB, ro); }
                    }
                    // log(q_i/P_i) = z_i ln2 - log(ssum) - log P_i: the softmax's own logits; the +1e-12 inside the reference's
                    // log and the q/(q+eps P) factor of its derivative change q*log(.) by < 1e-12, and the terms
                    // common to all levels cancel in the centred sum
                    v2f m2 = {0.f, 0.f}, m3 = {0.f, 0.f}, kk = {0.f, 0.f}, kl = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        const v2f d = amp[i] - m1, qd = q[i] * d, g = z[i] * LN2 + nlogP[i];
                        m2 += qd * d;
                        m3 += qd * d * d;
                        kk += qd * g;
                        kl += q[i] * g;
                    }
                    const bool inr0 = (n0 >= mh) && (n0 < B - mh) && act;             // KL slice, symbol index (:132)
                    const bool inr1 = (n0 + 1 >= mh) && (n0 + 1 < B - mh) && act;
                    if (inr0) klsum += kl.x - __builtin_amdgcn_logf(ssum.x) * LN2;
                    if (inr1) klsum += kl.y - __builtin_amdgcn_logf(ssum.y) * LN2;
                    asm volatile("" : "+v"(m2), "+v"(m3), "+v"(kk), "+v"(klsum));   // pin (see P1)
                    mv[0][o][c] = m2.x; mv[1][o][c] = m2.y;
                    mt3[0][o][c] = m3.x; mt3[1][o][c] = m3.y;
                    mkc[0][o][c] = inr0 ? kk.x : 0.f;
                    mkc[1][o][c] = inr1 ? kk.y : 0.f;
                    if (c) { muv[0].y = m1.x; muv[1].y = m1.y; } else { muv[0].x = m1.x; muv[1].x = m1.y; }
                    if (qf) {
#pragma unroll
                        for (int i = 0; i < NLEV; i++) {
#if VAEQ_ROW_STEP
                            const uint32_t ro = qrow;
#else
                            const uint32_t ro = (uint32_t)(o * 2 * NLEV + c * NLEV + i) * No4;
#endif
                            if (pairst) bst64(q[i], qr, vo0, ro);
                            else { bst32(q[i].x, qr, vo0, ro); bst32(q[i].y, qr, vo1, ro); }
#if VAEQ_ROW_STEP
                            qrow += No4;
                            asm volatile("" : "+s"(qrow));     // keeps it a running value (the optimiser would turn it back into products)
#endif
                        }
                    }
                }
                vv[o][0] = act ? mv[0][o][0] + mv[0][o][1] : 0.f;
                vv[o][1] = v1 ? mv[1][o][0] + mv[1][o][1] : 0.f;
                if (act) {                                     // U[nu=o][n]: even symbols in phase 0, odd in phase 1
                    Us[(o * 2 + 0) * Uph + gl] = muv[0];
                    Us[(o * 2 + 1) * Uph + gl] = v1 ? muv[1] : make_float2(0.f, 0.f);
                }
            }
            VAEQ_STAMP(4);
            // exclusive prefix sums PSv[nu][n], n = 0..B  (VS[nu][j] = PSv[hi+1] - PSv[lo])
            {
                float inc[2];
#pragma unroll
                for (int o = 0; o < 2; o++) inc[o] = wave_incl_scan_dpp(vv[o][0] + vv[o][1]);
                if constexpr (NW > 1) {                        // add the totals of the waves below (fixed order)
                    if (lane == 63) { RED[wv] = inc[0]; RED[NW + wv] = inc[1]; }
                    sync_lds<NW>();
#pragma unroll
                    for (int w = 0; w < NW - 1; w++)
                        if (w < wv) { inc[0] += RED[w]; inc[1] += RED[NW + w]; }
                }
#pragma unroll
                for (int o = 0; o < 2; o++) {
                    if (act) {
                        PSv[o * (BS + 1) + n0 + 1] = inc[o] - vv[o][1];
                        PSv[o * (BS + 1) + n0 + 2] = inc[o];
                    }
                    if (gl == 0) PSv[o * (BS + 1)] = 0.f;
                }
            }
            sync_lds<NW>();
            if (owner) {                                       // VS[nu][j]: lane = (j = tk, nu = half)
                const int lo = (Mh - tk + 1) >> 1, hi_ = (nm - 1 + Mh - tk) >> 1;
                VS[half * M + tk] = PSv[half * (BS + 1) + hi_ + 1] - PSv[half * (BS + 1) + lo];
            }
            sync_lds<NW>();

            VAEQ_STAMP(5);
            // ============ P3: residual e = x - D for the quad t = 4l'..4l'+3, both chi.
            //   D[chi, 2 tau + par] = sum_nu sum_a h[chi,nu,2a+par] U[nu, tau + mh - a],   tau in {2l', 2l'+1}, a = 0..mh
            float se0 = 0.f, se1 = 0.f;
            // |h|^2 and VS of this lane's (j, nu) for C and the G_V prefix sums: read now, so that their LDS latency hides behind the D loop
            float hq0 = 0.f, hq1 = 0.f;
            if (worker) {
                const float2 h0 = Ht[(0 * 2 + half) * MP + tk], h1 = Ht[(1 * 2 + half) * MP + tk];
                hq0 = h0.x * h0.x + h0.y * h0.y;
                hq1 = h1.x * h1.x + h1.y * h1.y;
            }
            const float vsl = worker ? VS[half * M + tk] : 0.f;
            {
                cacc D[2][4];                                  // [chi][i], i = 2*dl + par; lanes without a quad shadow lane 0 (no divergence, unused)
                constexpr int NA = mh + 1, NB = NA / 2;        // a = 0..mh; pairs (2b, 2b+1)
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    const float2 *h0 = Ht + (0 * 2 + v) * MP, *h1 = Ht + (1 * 2 + v) * MP;
                    const float2 *up = Us + (gl < nq ? gl : 0) + v * 2 * Uph;       // U[nu][2l' + d] = up[(d & 1) * Uph + (d >> 1)]
                    auto step_a = [&](int aa, v2f ulo, v2f uhi, auto first) {        // ulo = U[2l' + mh - a], uhi = U[2l' + mh - a + 1]
                        constexpr bool F = decltype(first)::value;
                        const v2f e0 = lds2(h0 + 2 * aa), o0 = lds2(h0 + 2 * aa + 1), e1 = lds2(h1 + 2 * aa), o1 = lds2(h1 + 2 * aa + 1);
                        cmacf<F>(D[0][0], e0, ulo); cmacf<F>(D[0][1], o0, ulo);
                        cmacf<F>(D[0][2], e0, uhi); cmacf<F>(D[0][3], o0, uhi);
                        cmacf<F>(D[1][0], e1, ulo); cmacf<F>(D[1][1], o1, ulo);
                        cmacf<F>(D[1][2], e1, uhi); cmacf<F>(D[1][3], o1, uhi);
                    };
                    // a = mh first (d = 0 -> U[2l'], U[2l'+1]; its odd tap is the zero pad when mh is even); for nu = 0 it starts the accumulators
                    if (NA & 1) {
                        if (v == 0) step_a(mh, lds2(up), lds2(up + Uph), std::true_type{});
                        else step_a(mh, lds2(up), lds2(up + Uph), std::false_type{});
                    } else if (v == 0) {
#pragma unroll
                        for (int chi = 0; chi < 2; chi++)
#pragma unroll
                            for (int i = 0; i < 4; i++) D[chi][i] = cacc0();
                    }
                    // stage b: a = 2b, 2b+1 (d = mh - 2b: samples d+1, d, d-1): 3 + 8 independent LDS reads feeding 32 packed FMAs
                    constexpr int ph = mh & 1;                 // phase of d (d and mh have equal parity)
                    auto load = [&](int b, v2f (&r)[11]) {
                        const int sl = (mh >> 1) - b;          // slot of d   (d >> 1)
                        r[0] = lds2(up + ph * Uph + sl);                            // d
                        r[1] = lds2(up + (ph ^ 1) * Uph + sl + ph);                 // d + 1
                        r[2] = lds2(up + (ph ^ 1) * Uph + sl + ph - 1);             // d - 1
#pragma unroll
                        for (int t = 0; t < 4; t++) { r[3 + t] = lds2(h0 + 4 * b + t); r[7 + t] = lds2(h1 + 4 * b + t); }   // (e, o) of a = 2b, 2b+1
                    };
                    auto fma = [&](int, const v2f (&r)[11], auto) {
                        cmac(D[0][0], r[3], r[0]); cmac(D[0][1], r[4], r[0]); cmac(D[0][2], r[3], r[1]); cmac(D[0][3], r[4], r[1]);
                        cmac(D[1][0], r[7], r[0]); cmac(D[1][1], r[8], r[0]); cmac(D[1][2], r[7], r[1]); cmac(D[1][3], r[8], r[1]);
                        cmac(D[0][0], r[5], r[2]); cmac(D[0][1], r[6], r[2]); cmac(D[0][2], r[5], r[0]); cmac(D[0][3], r[6], r[0]);
                        cmac(D[1][0], r[9], r[2]); cmac(D[1][1], r[10], r[2]); cmac(D[1][2], r[9], r[0]); cmac(D[1][3], r[10], r[0]);
                    };
                    pipe2<11, false, PIPE>(NB, load, fma);
                }
                VAEQ_STAMP(6);
                if (gl < nq) {
#pragma unroll
                    for (int chi = 0; chi < 2; chi++)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int ce = Mh + i;             // + 4*lane: same (phase, slot) arithmetic as x
                            const float2 x = Xl[(chi * 4 + (ce & 3)) * Lph + (ce >> 2)];
                            const float2 Dv = cfin(D[chi][i]);
                            float2 e = make_float2(x.x - Dv.x, x.y - Dv.y);
                            if (4 * gl + i >= nm) e = make_float2(0.f, 0.f);
                            Es[(chi * 4 + (ce & 3)) * Lph + gl + (ce >> 2)] = e;
                            const float e2 = e.x * e.x + e.y * e.y;
                            if (chi) se1 += e2; else se0 += e2;
                        }
                }
            }
            se0 = wave_sum_dpp(se0);
            se1 = wave_sum_dpp(se1);
            klsum = wave_sum_dpp(klsum);
            if constexpr (NW > 1) {                            // totals over the run's waves, same order in every wave
                if (lane == 0) { RED[16 + wv] = se0; RED[16 + NW + wv] = se1; RED[16 + 2 * NW + wv] = klsum; }
                sync_lds<NW>();
                se0 = RED[16]; se1 = RED[16 + NW]; klsum = RED[16 + 2 * NW];
#pragma unroll
                for (int w = 1; w < NW; w++) { se0 += RED[16 + w]; se1 += RED[16 + NW + w]; klsum += RED[16 + 2 * NW + w]; }
            }
            // C[chi] = sum|e|^2 + sum_{nu,j} |h|^2 VS   (lanes (j, nu) hold one term each for both chi; hq, vsl were read before the D loop)
            const float C0 = se0 + wave_sum_dpp(hq0 * vsl), C1 = se1 + wave_sum_dpp(hq1 * vsl);
            // C is uniform: hardware reciprocal / log2 (1 ulp: 6e-8 on the gradients' common scale, 1e-7 relative on the ELBO) instead of the IEEE
            // division and logf expansions (~60 instructions per step that every lane would execute for lane 0's two stores)
            const float gC0 = (float)nm * __builtin_amdgcn_rcpf(C0), gC1 = (float)nm * __builtin_amdgcn_rcpf(C1);
            {
                const uint32_t vo = gl == 0 ? 0u : OOB;        // lane 0 stores; row offsets ride in the scalar offset
                if (a.loss) bst32((float)nm * LN2 * (__builtin_amdgcn_logf(C0) + __builtin_amdgcn_logf(C1)) + klsum, lr_, vo, (uint32_t)s * 4u);
                if (a.var_est) {
                    bst32(C0 * rnm, vr_, vo, (uint32_t)s * 4u);
                    bst32(C1 * rnm, vr_, vo, ((uint32_t)a.steps + (uint32_t)s) * 4u);
                }
            }
            // prefix sums over j of H2[nu][j] = sum_chi gC[chi] |h[chi,nu,j]|^2  -> G_V by two lookups per symbol
            {
                const float inc = half_incl_scan_dpp(gC0 * hq0 + gC1 * hq1);   // inclusive scan within each 32-lane half
                if (owner) PSh[half * MP + tk + 1] = inc;
                if (tk == 0) PSh[half * MP] = 0.f;
            }
            sync_lds<NW>();

            VAEQ_STAMP(7);
            // ============ P4a: dL/dh partial sums, lane = (j = tk, half of the tau range); acc[chi][nu]
            step += 1;
            b1t *= 0.9;
            b2t *= 0.999;
            const float rbc1 = __builtin_amdgcn_rcpf((float)(1.0 - b1t));                 // bias corrections: beta^t in double,
            const float bc2s = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((float)(1.0 - b2t)));   // the rest in float
            const float ssW = lrW * rbc1, ssH = lrH * rbc1;
            float2 hnew[2];
            hnew[0] = hnew[1] = make_float2(0.f, 0.f);
            float ghr[2] = {0, 0}, ghi[2] = {0, 0};
            float2 hacc[2];                                    // NW > 1: this wave's part of sum e conj(U) for (chi = half, nu)
            {
                // what the owner lane's Adam(h) update reads of the taps and VS is fetched from LDS BEFORE the tap loop, so that the
                // update after the loop starts with fewer exposed LDS round trips (wave 0 only)
                float2 ph_h[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
                float ph_vs[2] = {0.f, 0.f};                   // (the moments stay in LDS until the update: this phase cannot spare eight more registers)
                if (NW == 1 && owner) {
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        ph_h[v] = Ht[(half * 2 + v) * MP + tk];
                        ph_vs[v] = VS[v * M + tk];
                    }
                }
                cacc ca[2][2];
                {
                    // sum over tau of e[chi, 2 tau + par] conj(U[nu, tau + mh - a]); tau runs in pairs (2m, 2m+1) so that the
                    // polyphase component of every operand is a per-lane constant and only the slot advances (by one per m).
                    // Past the last valid tau the residual cells are zero (pad), so the pair loop may overrun by one.
                    // Lanes beyond the M taps shadow tap 0 (no divergence; their sums are never read).
                    const int tkc = worker ? tk : 0, par = tkc & 1, aa = tkc >> 1;
                    const int T = (nm - par + 1) >> 1, Th = ((T + 2 * NP - 1) / (2 * NP)) << 1;   // even split points
                    const int ma = (part * Th) >> 1, mb = (min(T, part * Th + Th) + 1) >> 1;
                    const int ceA = par + Mh, ceB = par + Mh + 2, npA = mh - aa, npB = mh - aa + 1;
                    const float2 *eA = Es + (ceA & 3) * Lph + (ceA >> 2) + ma, *eB = Es + (ceB & 3) * Lph + (ceB >> 2) + ma;
                    const float2 *uA = Us + (npA & 1) * Uph + (npA >> 1) + ma, *uB = Us + (npB & 1) * Uph + (npB >> 1) + ma;
                    // stage = two pairs (m = 2i, 2i+1): 16 independent LDS reads feeding 32 packed FMAs (a one-pair stage's 16 FMAs are too short
                    // to cover the next stage's LDS latency); an odd last pair is handled after the pipeline
                    auto load1 = [&](int m, v2f *r) {
                        r[0] = lds2(eA + m); r[1] = lds2(eA + 4 * Lph + m); r[2] = lds2(uA + m); r[3] = lds2(uA + 2 * Uph + m);
                        r[4] = lds2(eB + m); r[5] = lds2(eB + 4 * Lph + m); r[6] = lds2(uB + m); r[7] = lds2(uB + 2 * Uph + m);
                    };
                    cacc cbx[2][2];                            // the odd tau of every pair sum here: 16 independent chains (see pair_fir) (WIDE)
                    cacc (&cb)[2][2] = WIDE ? cbx : ca;
                    auto fma1 = [&](const v2f *r, auto first) {
                        constexpr bool F = decltype(first)::value;
                        cmacf<F>(ca[0][0], r[2], r[0]); cmacf<F>(ca[0][1], r[3], r[0]); cmacf<F>(ca[1][0], r[2], r[1]); cmacf<F>(ca[1][1], r[3], r[1]);
                        cmacf<F && WIDE>(cb[0][0], r[6], r[4]); cmacf<F && WIDE>(cb[0][1], r[7], r[4]); cmacf<F && WIDE>(cb[1][0], r[6], r[5]); cmacf<F && WIDE>(cb[1][1], r[7], r[5]);
                    };
                    auto load = [&](int i, v2f (&r)[16]) { load1(2 * i, r); load1(2 * i + 1, r + 8); };
                    auto fma = [&](int, const v2f (&r)[16], auto first) { fma1(r, first); fma1(r + 8, std::false_type{}); };
                    int n = mb - ma;
                    VAEQ_XSTAMP(0);
                    if constexpr (UNI) {                       // baked shape: every part has the same number (>= 2) of pairs
                        n = __builtin_amdgcn_readfirstlane(n);
                        pipe2<16, true, PIPE>(n >> 1, load, fma);
                    } else {
                        ca[0][0] = ca[0][1] = ca[1][0] = ca[1][1] = cbx[0][0] = cbx[0][1] = cbx[1][0] = cbx[1][1] = cacc0();
                        pipe2<16, false, PIPE>(n >> 1, load, fma);
                    }
                    if (n & 1) {
                        v2f r[8];
                        load1(n - 1, r);
                        fma1(r, std::false_type{});
                    }
                    if constexpr (WIDE) {
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++) { ca[i][j].a += cbx[i][j].a; ca[i][j].b += cbx[i][j].b; }
                    }
                }
                VAEQ_XSTAMP(1);
                float2 acc[2][2];
#pragma unroll
                for (int chi = 0; chi < 2; chi++)
#pragma unroll
                    for (int v = 0; v < 2; v++) acc[chi][v] = cfinc(ca[chi][v]);             // e * conj(U)
                // combine the two halves; lane (j, half) keeps chi = half
#pragma unroll
                for (int chi = 0; chi < 2; chi++)
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        acc[chi][v].x += __shfl_xor(acc[chi][v].x, 32, 64);
                        acc[chi][v].y += __shfl_xor(acc[chi][v].y, 32, 64);
                    }
                if constexpr (NW > 1) {                        // waves 1.. hand their partial sums to wave 0 (read after the next barrier)
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        hacc[v] = half ? acc[1][v] : acc[0][v];
                        if (wv > 0) {
                            XG[((wv - 1) * 4 + 2 * v + 0) * 64 + lane] = hacc[v].x;
                            XG[((wv - 1) * 4 + 2 * v + 1) * 64 + lane] = hacc[v].y;
                        }
                    }
                }
                VAEQ_XSTAMP(2);
                if (NW == 1 && owner) {
                    const float g = half ? gC1 : gC0;
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const float2 ac = half ? acc[1][v] : acc[0][v];
                        const float2 hh = ph_h[v];
                        const float vs = ph_vs[v];
                        ghr[v] = g * (-2.0f * ac.x + 2.0f * hh.x * vs);
                        ghi[v] = g * (-2.0f * ac.y + 2.0f * hh.y * vs);
                        hnew[v] = hh;
                        if (!a.no_update) {
                            adam_update_fast(hnew[v].x, mHr(v), vHr(v), ghr[v], ssH, bc2s);
                            adam_update_fast(hnew[v].y, mHi(v), vHi(v), ghi[v], ssH, bc2s);
                        }
                    }
                }
            }

            asm volatile("" : "+v"(hnew[0].x), "+v"(hnew[0].y), "+v"(hnew[1].x), "+v"(hnew[1].y), "+v"(ghr[0]), "+v"(ghr[1]), "+v"(ghi[0]),
                         "+v"(ghi[1]));                        // pin (see P1)
            // prefetch the next window now: the q/y stores of this step were issued half a step ago and have drained, so the wait
            // for these loads at the top of the next step does not also wait for fresh stores (vmcnt retires in order)
            {
                const bool last_s = s + 1 == a.steps;
                if (!(last_s && f + 1 == a.n_frames)) fetch(last_s ? xn : xr, last_s ? 0 : s + 1);
            }
            VAEQ_XSTAMP(3);
            VAEQ_STAMP(8);
            // ============ P4b: dL/dU for the lane's symbol pair (same shape as the FIR, on e with conj(h)), then dL/dy
            float2 gy[2][2];                                   // [sym][nu]
            {
                float2 au0[2][2], au1[2][2];                   // chi = 0 / 1: [sym][nu]
                {
                    cacc cu[2][2][2];                          // [chi][sym][nu]
                    pair_du<M, PIPE_DU, WIDE>(cu, Ea, Lph, Ht, MP);
#pragma unroll
                    for (int sy = 0; sy < 2; sy++)
#pragma unroll
                        for (int v = 0; v < 2; v++) { au0[sy][v] = cfinc(cu[0][sy][v]); au1[sy][v] = cfinc(cu[1][sy][v]); }   // e * conj(h)
                }
#pragma unroll
                for (int sy = 0; sy < 2; sy++) {
                    const int sx = 2 * (n0 + sy);
                    const int jlo = max(0, Mh - sx), jhi = max(jlo - 1, min(Mh, nm - 1 + Mh - sx));
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const float ur = -2.0f * (gC0 * au0[sy][v].x + gC1 * au1[sy][v].x);
                        const float ui = -2.0f * (gC0 * au0[sy][v].y + gC1 * au1[sy][v].y);
                        const float gv = PSh[v * MP + jhi + 1] - PSh[v * MP + jlo];
                        const float iv = 1.0f / (v ? var1 : var0);
                        gy[sy][v].x = iv * (ur * mv[sy][v][0] + gv * mt3[sy][v][0] + mkc[sy][v][0]);
                        gy[sy][v].y = iv * (ui * mv[sy][v][1] + gv * mt3[sy][v][1] + mkc[sy][v][1]);
                    }
                }
            }
            VAEQ_STAMP(9);
            sync_lds<NW>();                                   // every read of U / old h is done (GY aliases U)
            if constexpr (NW > 1) {
                if (owner) {
                    const float g = half ? gC1 : gC0;
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        float2 ac = hacc[v];
#pragma unroll
                        for (int w = 0; w < NW - 1; w++) {
                            ac.x += XG[(w * 4 + 2 * v + 0) * 64 + lane];
                            ac.y += XG[(w * 4 + 2 * v + 1) * 64 + lane];
                        }
                        const float2 hh = Ht[(half * 2 + v) * MP + tk];
                        const float vs = VS[v * M + tk];
                        ghr[v] = g * (-2.0f * ac.x + 2.0f * hh.x * vs);
                        ghi[v] = g * (-2.0f * ac.y + 2.0f * hh.y * vs);
                        hnew[v] = hh;
                        if (!a.no_update) {
                            adam_update_fast(hnew[v].x, mHr(v), vHr(v), ghr[v], ssH, bc2s);
                            adam_update_fast(hnew[v].y, mHi(v), vHi(v), ghi[v], ssH, bc2s);
                        }
                    }
                }
            }
            if (act) {
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    GY[v * BS + n0] = gy[0][v];
                    GY[v * BS + n0 + 1] = v1 ? gy[1][v] : make_float2(0.f, 0.f);
                }
            }
            if (owner && !a.no_update) {
                Ht[(half * 2 + 0) * MP + tk] = hnew[0];
                Ht[(half * 2 + 1) * MP + tk] = hnew[1];
            }
            sync_lds<NW>();

            VAEQ_STAMP(10);
            // ============ P5: dL/dw partial sums, lane = (k = tk, half of the symbol range); acc[o][p]
            float gwr[2] = {0, 0}, gwi[2] = {0, 0};
            {
                float pw_w[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, pw_m[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, pw_v[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // as for Adam(h)
                if (owner) {
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const float *wq = reinterpret_cast<const float *>(&Wt[p * M + tk]) + half * 2;
                        pw_w[p][0] = wq[0]; pw_w[p][1] = wq[1];
                        pw_m[p][0] = mWr(p); pw_m[p][1] = mWi(p);
                        pw_v[p][0] = vWr(p); pw_v[p][1] = vWi(p);
                    }
                }
                cacc ca[2][2];
                {
                    // sum over n of gy[o, n] conj(x[p, 2n + k]); n runs in pairs (2m, 2m+1): x phase fixed per lane, gy pair = 16 bytes
                    const int tkc = worker ? tk : 0;           // lanes beyond the M taps shadow tap 0
                    const int Be = ODDB ? B + (B & 1) : B;     // odd B: the last pair's second symbol has dL/dy = 0
                    const int Bq = ((Be + 2 * NP - 1) / (2 * NP)) << 1;                  // even split points
                    const int ma = (part * Bq) >> 1, mb = min(Be, part * Bq + Bq) >> 1;
                    const int cA = tkc, cB = tkc + 2;
                    const float2 *xA = Xs + (cA & 3) * Lph + (cA >> 2) + ma, *xB = Xs + (cB & 3) * Lph + (cB >> 2) + ma;
                    const float2 *G0 = GY + 2 * ma, *G1 = GY + BS + 2 * ma;
                    auto load1 = [&](int m, v2f *r) {
                        r[0] = lds2(G0 + 2 * m); r[1] = lds2(G0 + 2 * m + 1); r[2] = lds2(G1 + 2 * m); r[3] = lds2(G1 + 2 * m + 1);   // gy[o][2m], gy[o][2m+1]
                        r[4] = lds2(xA + m); r[5] = lds2(xA + 4 * Lph + m); r[6] = lds2(xB + m); r[7] = lds2(xB + 4 * Lph + m);
                    };
                    cacc cbx[2][2];                            // the odd symbols of every pair sum here: 16 independent chains (WIDE)
                    cacc (&cb)[2][2] = WIDE ? cbx : ca;
                    auto fma1 = [&](const v2f *r, auto first) {
                        constexpr bool F = decltype(first)::value;
                        cmacf<F>(ca[0][0], r[4], r[0]); cmacf<F>(ca[0][1], r[5], r[0]); cmacf<F>(ca[1][0], r[4], r[2]); cmacf<F>(ca[1][1], r[5], r[2]);
                        cmacf<F && WIDE>(cb[0][0], r[6], r[1]); cmacf<F && WIDE>(cb[0][1], r[7], r[1]); cmacf<F && WIDE>(cb[1][0], r[6], r[3]); cmacf<F && WIDE>(cb[1][1], r[7], r[3]);
                    };
                    auto load = [&](int i, v2f (&r)[16]) { load1(2 * i, r); load1(2 * i + 1, r + 8); };       // stage = two symbol pairs (see dL/dh)
                    auto fma = [&](int, const v2f (&r)[16], auto first) { fma1(r, first); fma1(r + 8, std::false_type{}); };
                    int n = mb - ma;
                    VAEQ_XSTAMP(4);
                    if constexpr (UNI) {
                        n = __builtin_amdgcn_readfirstlane(n);
                        pipe2<16, true, PIPE>(n >> 1, load, fma);
                    } else {
                        ca[0][0] = ca[0][1] = ca[1][0] = ca[1][1] = cbx[0][0] = cbx[0][1] = cbx[1][0] = cbx[1][1] = cacc0();
                        pipe2<16, false, PIPE>(n >> 1, load, fma);
                    }
                    if (n & 1) {
                        v2f r[8];
                        load1(n - 1, r);
                        fma1(r, std::false_type{});
                    }
                    if constexpr (WIDE) {
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++) { ca[i][j].a += cbx[i][j].a; ca[i][j].b += cbx[i][j].b; }
                    }
                }
                VAEQ_XSTAMP(5);
                float2 acc[2][2];
#pragma unroll
                for (int o = 0; o < 2; o++)
#pragma unroll
                    for (int pp = 0; pp < 2; pp++) acc[o][pp] = cfinc(ca[o][pp]);            // gy * conj(x)
#pragma unroll
                for (int o = 0; o < 2; o++)
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        acc[o][p].x += __shfl_xor(acc[o][p].x, 32, 64);
                        acc[o][p].y += __shfl_xor(acc[o][p].y, 32, 64);
                    }
                if constexpr (NW > 1) {                        // as for dL/dh: wave 0 adds the other waves' parts after the barrier
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const float2 ac = half ? acc[1][p] : acc[0][p];
                        if (wv > 0) {
                            XG[((wv - 1) * 4 + 2 * p + 0) * 64 + lane] = ac.x;
                            XG[((wv - 1) * 4 + 2 * p + 1) * 64 + lane] = ac.y;
                        }
                    }
                    sync_lds<NW>();
                }
                VAEQ_XSTAMP(6);
                if (owner) {
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        float2 ac = half ? acc[1][p] : acc[0][p];
                        if constexpr (NW > 1) {
#pragma unroll
                            for (int w = 0; w < NW - 1; w++) {
                                ac.x += XG[(w * 4 + 2 * p + 0) * 64 + lane];
                                ac.y += XG[(w * 4 + 2 * p + 1) * 64 + lane];
                            }
                        }
                        gwr[p] = ac.x;
                        gwi[p] = ac.y;
                        if (!a.no_update) {
                            float *wq = reinterpret_cast<float *>(&Wt[p * M + tk]) + half * 2;
                            float wr = pw_w[p][0], wi = pw_w[p][1];
                            adam_update_fast(wr, pw_m[p][0], pw_v[p][0], gwr[p], ssW, bc2s);
                            adam_update_fast(wi, pw_m[p][1], pw_v[p][1], gwi[p], ssW, bc2s);
                            wq[0] = wr;
                            wq[1] = wi;
                            mWr(p) = pw_m[p][0]; mWi(p) = pw_m[p][1];
                            vWr(p) = pw_v[p][0]; vWi(p) = pw_v[p][1];
                        }
                    }
                }
            }
            if (a.dbg_gW && owner && f == a.n_frames - 1 && s == a.steps - 1) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    a.dbg_gW[gbase + (half * 4 + p) * M + tk] = gwr[p];
                    a.dbg_gW[gbase + (half * 4 + 2 + p) * M + tk] = gwi[p];
                    a.dbg_gh[gbase + ((half * 2 + p) * 2 + 0) * M + tk] = ghr[p];
                    a.dbg_gh[gbase + ((half * 2 + p) * 2 + 1) * M + tk] = ghi[p];
                }
            }
            sync_lds<NW>();
            VAEQ_XSTAMP(7);
            VAEQ_STAMP(11);
        }
    }
#ifdef VAEQ_PHASE_STAMPS
    if (gl == 0 && run == (int)gridDim.x / 2 && a.loss)
        for (int i = 0; i + 1 < VAEQ_NSTAMP; i++) a.loss[i] = (float)(tst[i + 1] - tst[i]);
    if (gl == 0 && run == (int)gridDim.x / 2 && a.loss)
        for (int i = 0; i < 7; i++) a.loss[16 + i] = (float)(tsx[i + 1] - tsx[i]);
#endif

    // ---- state out
    if (owner && !a.no_update) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const size_t ir = gbase + (half * 4 + p) * M + tk, ii = gbase + (half * 4 + 2 + p) * M + tk;
            const float *wq = reinterpret_cast<const float *>(&Wt[p * M + tk]) + half * 2;
            a.W[ir] = wq[0]; a.W[ii] = wq[1];
            a.adam_mW[ir] = mWr(p); a.adam_mW[ii] = mWi(p);
            a.adam_vW[ir] = vWr(p); a.adam_vW[ii] = vWi(p);
            const size_t hr = gbase + ((half * 2 + p) * 2 + 0) * M + tk, hi = hr + M;
            const float2 hh = Ht[(half * 2 + p) * MP + tk];
            a.h[hr] = hh.x; a.h[hi] = hh.y;
            a.adam_mh[hr] = mHr(p); a.adam_mh[hi] = mHi(p);
            a.adam_vh[hr] = vHr(p); a.adam_vh[hi] = vHi(p);
        }
    }
    if (gl == 0 && !a.no_update) a.step[run] = step;
#undef mWr
#undef mWi
#undef vWr
#undef vWi
#undef mHr
#undef mHi
#undef vHr
#undef vHi
}


template <int M, int NLEV, int BT, int NW>
static int launch_wave(const vaeq_dp_args &a, hipStream_t st)
{
    const size_t lds = (size_t)wave_layout(a.B, M, NW).total;
    const bool pair = ((a.keep_len | a.keep_off) & 1) == 0;
    void (*k)(const vaeq_dp_args) = pair ? dp_wave_kernel<M, NLEV, BT, true, 0, NW> : dp_wave_kernel<M, NLEV, BT, false, 0, NW>;
    if (BT) {                                                  // the tuned shape also gets the output-mode specialisations
        if (!a.eq_out && !a.dec_out) k = pair ? dp_wave_kernel<M, NLEV, BT, true, BT ? 1 : 0, NW> : dp_wave_kernel<M, NLEV, BT, false, BT ? 1 : 0, NW>;
        else if (!a.q_out) k = pair ? dp_wave_kernel<M, NLEV, BT, true, BT ? 2 : 0, NW> : dp_wave_kernel<M, NLEV, BT, false, BT ? 2 : 0, NW>;
    }
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    const int out = !BT ? 0 : (!a.eq_out && !a.dec_out) ? 1 : !a.q_out ? 2 : 0;
    note_kernel("vaeq::dp_wave_kernel<%d, %d, %d, %s, %d, %d, 0>", M, NLEV, BT, pair ? "true" : "false", out, NW);   // every template argument, as rocprofv3 prints the name
    hipLaunchKernelGGL(k, dim3(a.R), dim3(64 * NW), lds, st, a);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

// run-time B <= BL on the fixed LDS layout of BL (immediate offsets and strides, pipelined tap loops), with the output-mode specialisations
// (SPEC = false: only the all-outputs-nullable instantiation, for the tap counts that are not the reference's default)
template <int M, int NLEV, int BL, int NW, bool SPEC = true>
static int launch_wave_fixl(const vaeq_dp_args &a, hipStream_t st)
{
    const size_t lds = (size_t)wave_layout(BL, M, NW).total;
    const bool pair = ((a.keep_len | a.keep_off) & 1) == 0;
    const int out = !SPEC ? 0 : (!a.eq_out && !a.dec_out) ? 1 : !a.q_out ? 2 : 0;
    void (*k)(const vaeq_dp_args) = pair ? dp_wave_kernel<M, NLEV, 0, true, 0, NW, BL> : dp_wave_kernel<M, NLEV, 0, false, 0, NW, BL>;
    if constexpr (SPEC) {
        if (out == 1) k = pair ? dp_wave_kernel<M, NLEV, 0, true, 1, NW, BL> : dp_wave_kernel<M, NLEV, 0, false, 1, NW, BL>;
        else if (out == 2) k = pair ? dp_wave_kernel<M, NLEV, 0, true, 2, NW, BL> : dp_wave_kernel<M, NLEV, 0, false, 2, NW, BL>;
    }
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    note_kernel("vaeq::dp_wave_kernel<%d, %d, 0, %s, %d, %d, %d>", M, NLEV, pair ? "true" : "false", out, NW, BL);
    hipLaunchKernelGGL(k, dim3(a.R), dim3(64 * NW), lds, st, a);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

template <int M, int NLEV, int BL, int NW>
static int64_t wave_resident_fixl()
{
    int nb = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return VAEQ_ERR_DEVICE;
    auto k = dp_wave_kernel<M, NLEV, 0, true, 0, NW, BL>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64 * NW, (size_t)wave_layout(BL, M, NW).total) != hipSuccess) return VAEQ_ERR_DEVICE;
    return (int64_t)nb * prop.multiProcessorCount;
}

template <int M, int BT, int NW>
static int launch_wave_lev(const vaeq_dp_args &a, hipStream_t st)
{
    switch (a.n_lev) {
    case 2: return launch_wave<M, 2, BT, NW>(a, st);
    case 4: return launch_wave<M, 4, BT, NW>(a, st);
    case 8: return launch_wave<M, 8, BT, NW>(a, st);
    }
    return VAEQ_ERR_SHAPE;
}

template <int M, int NLEV, int BT, int NW>
static int64_t wave_resident(int B)
{
    int nb = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return VAEQ_ERR_DEVICE;
    const size_t lds = (size_t)wave_layout(B, M, NW).total;
    auto k = dp_wave_kernel<M, NLEV, BT, true, 0, NW>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64 * NW, lds) != hipSuccess) return VAEQ_ERR_DEVICE;
    return (int64_t)nb * prop.multiProcessorCount;
}

template <int M, int BT, int NW>
static int64_t wave_resident_lev(int B, int n_lev)
{
    switch (n_lev) {
    case 2: return wave_resident<M, 2, BT, NW>(B);
    case 4: return wave_resident<M, 4, BT, NW>(B);
    case 8: return wave_resident<M, 8, BT, NW>(B);
    }
    return VAEQ_ERR_SHAPE;
}

// every supported M for a given NW (the per-NW translation units instantiate these): M = 25 on the run-time layout (its fixed-layout and baked forms
// are dispatched before; this is their A/B counterpart), every other tap count on the fixed layout of the NW class's largest minibatch
template <int M, int NW>
static int launch_wave_fixl_lev(const vaeq_dp_args &a, hipStream_t st)
{
    switch (a.n_lev) {
    case 2: return launch_wave_fixl<M, 2, 128 * NW, NW, false>(a, st);
    case 4: return launch_wave_fixl<M, 4, 128 * NW, NW, false>(a, st);
    case 8: return launch_wave_fixl<M, 8, 128 * NW, NW, false>(a, st);
    }
    return VAEQ_ERR_SHAPE;
}
template <int M, int NW>
static int64_t wave_resident_fixl_lev(int n_lev)
{
    switch (n_lev) {
    case 2: return wave_resident_fixl<M, 2, 128 * NW, NW>();
    case 4: return wave_resident_fixl<M, 4, 128 * NW, NW>();
    case 8: return wave_resident_fixl<M, 8, 128 * NW, NW>();
    }
    return VAEQ_ERR_SHAPE;
}

template <int NW>
static int launch_wave_any(const vaeq_dp_args &a, hipStream_t st)
{
    switch (a.M) {
    case 25: return launch_wave_lev<25, 0, NW>(a, st);
    case 31: return launch_wave_fixl_lev<31, NW>(a, st);
    case 21: return launch_wave_fixl_lev<21, NW>(a, st);
    case 17: return launch_wave_fixl_lev<17, NW>(a, st);
    case 13: return launch_wave_fixl_lev<13, NW>(a, st);
    case 9: return launch_wave_fixl_lev<9, NW>(a, st);
    }
    return VAEQ_ERR_SHAPE;
}

template <int NW>
static int64_t wave_resident_any(int B, int M, int n_lev)
{
    switch (M) {
    case 25: return wave_resident_lev<25, 0, NW>(B, n_lev);
    case 31: return wave_resident_fixl_lev<31, NW>(n_lev);
    case 21: return wave_resident_fixl_lev<21, NW>(n_lev);
    case 17: return wave_resident_fixl_lev<17, NW>(n_lev);
    case 13: return wave_resident_fixl_lev<13, NW>(n_lev);
    case 9: return wave_resident_fixl_lev<9, NW>(n_lev);
    }
    return VAEQ_ERR_SHAPE;
}

}  // namespace vaeq
