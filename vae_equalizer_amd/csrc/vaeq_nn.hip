// vaeq_nn.hip -- SURVEY row f3: the AWGN VAE-NN equalizer (AWGN_channel/func_VAENN_MQAM.py), training loop and validation pass.
//
//   Net.forward (:170-188)   fc1 = Conv1d(2, C, k1, pad k1/2) -> ELU -> fc2 = Conv1d(C, C, k2, pad k2/2, stride sps), C = 2 n_lev;
//                            per-axis softmax over the n_lev logits.  The residual x_res (:181-186) adds the SAME number to every
//                            logit of an axis, which a softmax cancels exactly -- it is not computed here.
//   loss_function (:63-95)   the VAE-LE ELBO of the AWGN channel with the entropy of q in place of the KL to a prior:
//                            loss = nm log C + sum q log(q + 1e-12),  C = sum|x - D|^2 + sum_j |h_j|^2 VS[j]
//   Adam(amsgrad=True) (:248-253) on every parameter, one learning rate.
// Parameters of a run are ONE flat vector theta = [fc1.weight C*2*k1 | fc1.bias C | fc2.weight C*C*k2 | fc2.bias C | h_est 2*M]
// (the order of net.parameters() followed by h_est); gradients and the three Adam vectors use the same layout.
//
// One workgroup per run, everything in LDS (105 KB for 64-QAM / k1 = 25 / B = 300): x window, ELU output z1 (C x L, reused in
// place for dL/dz1), logits -> q -> dL/dlogits in place, the loss intermediates of vaeq_awgn.hip, theta, gradient, Adam state.
// Backward = the closed form of the ELBO (vaeq_dp.hip) down to dL/dq, then softmax / conv / ELU / conv backward by the chain rule.
// Forward convolutions: item = (4 output channels, sample), consecutive lanes on consecutive samples, the 4 channels' weights of a
// tap in one 16-byte broadcast read of a transposed weight copy.  Weight gradients: one wave per (input, tap) column, lanes stride
// over the samples with all C channel sums in registers, then wave reductions -- every LDS access pattern is conflict free.
// C = 16 (64-QAM): these five GEMM-shaped phases run on v_mfma_f32_16x16x4_f32 instead (mfma_conv16 / mfma_wgrad16 / mfma_convT16 below).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vaeq.h"
#include "vaeq_common.h"
#include "vaeq_validate.h"

#ifndef VAEQ_NN_PREF
#define VAEQ_NN_PREF 0                                 // 1: baked 64-QAM `Net` kernel fetches the next minibatch into registers during the step (measured: no gain, 16 spilled registers)
#endif
#ifndef VAEQ_NN_LEAN
#define VAEQ_NN_LEAN 0                                 // 1: ... its forward convolutions with the k-step loop kept a loop (measured: -4 %)
#endif

#ifndef VAEQ_NN_MFMA8
#define VAEQ_NN_MFMA8 1                                // 16-QAM (8 channels) on the 16-row MFMA path of 64-QAM, rows 8..15 zero (0: the vector-ALU path)
#endif

namespace vaeq {

// which alphabets run the convolutions on v_mfma_f32_16x16x4_f32, and the channel rows their LDS buffers carry (MFMA rows: 16)
__host__ __device__ constexpr bool nn_mf(int n) { return n == 8 || (n == 4 && VAEQ_NN_MFMA8); }
__host__ __device__ constexpr int nn_cp(int n) { return nn_mf(n) ? 16 : 2 * n; }

struct NNLayout {
    int C, L, p1, p2, Lx, Lz, mh, Mh, nm, NP, NW1, oW1, oB1, oW2, oB2, oG, oBt, oH;
    int AS, A0;                                        // row stride / first column of a2 (training, C = 16: zero guard columns around the B logits of a row)
    int ES, PH;                                        // row stride of the zero-guarded residual rows; offset of the |h|^2 prefix sums
    int xs, z1, zb, bnst, a2, mu, vr, es, VS, th, gr, am, av, ax, w1t, w2t, w2u, red, total;
};

__host__ __device__ inline int npad4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline NNLayout nn_layout(int B, int sps, int M, int n, int k1, int k2, bool bn = false, bool eval = false)
{
    NNLayout l;
    l.C = 2 * n; l.L = B * sps; l.p1 = k1 / 2; l.p2 = k2 / 2;
    l.Lx = npad4(l.L + 2 * l.p1 + 8);                  // zero halo + room for the 4-wide windows of the last quad
    l.Lz = npad4(l.L + 2 * l.p2 + 4);
    const bool mf = nn_mf(n);
    const int CP = nn_cp(n);                           // channel rows in LDS (MFMA path: 16, the rows past C stay zero)
    if (mf)                                            // MFMA path: row stride an odd multiple of 4 dwords, so that 16 channels x 4
        while ((l.Lz & 7) != 4) l.Lz += 4;             // consecutive samples (an MFMA operand / result) fall into 64 different banks
    l.mh = M / 2; l.Mh = 2 * l.mh; l.nm = l.L - l.Mh;
    l.NW1 = l.C * 2 * k1;
    l.oW1 = 0; l.oB1 = l.NW1; l.oW2 = l.oB1 + l.C; l.oB2 = l.oW2 + l.C * l.C * k2;
    l.oG = l.oB2 + l.C; l.oBt = l.oG + l.C;            // BatchNorm weight / bias (Net_BN only)
    l.oH = bn ? l.oBt + l.C : l.oG; l.NP = l.oH + 2 * M;
    int o = 0;
    auto take = [&](int cnt) { int r = o; o += npad4(cnt); return r; };
    const int one = (mf && !eval) ? 1 : 0;         // training on the MFMA path: a row of ones behind the input rows and behind the channel rows (the
                                                       // bias columns of the weight-gradient GEMMs read it like any other operand row: mfma_wgrad16, ROW1)
    l.xs = take((2 + one) * l.Lx);
    l.z1 = take((CP + (bn ? 0 : one)) * l.Lz);         // (Net_BN: fc2's input, and with it the row of ones, is zb)
    l.zb = bn && !eval ? take((CP + one) * l.Lz) : l.z1;      // Net_BN: BatchNorm output (fc2's input); z1 then holds the normalised zhat
                                                       // (eval mode folds the running statistics into fc1's epilogue: no second buffer)
    l.bnst = take(bn ? 6 * l.C : 0);                   // mean, rstd (batch) | running_mean, running_var | eval scale, shift
    // training on the MFMA path: the backward pass through fc2 reads dL/dlogits at n + shift, shift in [-4, 4], for whole 16-column tiles -- with
    // A0 zero columns in front, the row padded to whole tiles + A0 behind and a stride = 4 (mod 8) (conflict-free MFMA operand reads) no read needs a
    // clamp or a condition (a conditional LDS read costs a branch and an exposed round trip each: mfma_convT16)
    l.A0 = (mf && !eval) ? 4 : 0;
    l.AS = B;
    if (mf && !eval) {
        l.AS = 16 * ((B + 15) / 16) + 2 * l.A0;
        while ((l.AS & 7) != 4) l.AS += 4;
    }
    l.a2 = take(CP * l.AS + 2 * l.A0);
    l.mu = take(2 * B); l.vr = take(2 * B);
    l.ES = npad4(l.nm + 2 * l.Mh + 4);                 // Mh zeros | nm residual samples | Mh + 4 zeros (nn_train_kernel)
    l.es = take(2 * l.ES);
    l.VS = take(M);
    l.PH = take(M + 1);
    l.th = take(l.NP);
    const int NPt = eval ? 0 : l.NP;                   // gradient and AMSGrad state: training only
    l.gr = take(NPt); l.am = take(NPt); l.av = take(NPt); l.ax = take(NPt);
    // transposed weight copies: MFMA path = the walk order of mfma_conv16 (16 channel columns, 64 floats per k-step), else [i][k][c] / [cc][k][c]
    l.w1t = take(mf ? 64 * ((((k1 + 1) / 2) + 1) & ~1) : l.NW1 + 7 * l.C);
    l.w2t = take(mf ? 256 * k2 : l.C * l.C * k2 + 7 * l.C);
    l.w2u = take(eval ? 0 : CP * CP * k2);             // fc2.weight as [k][c][cc] (backward through fc2)
    l.red = take(64);
    l.total = o;
    return l;
}

// Sum C per-lane partials over the 64 lanes of a wave, all C at once: log2(C) halving rounds (a lane hands over half of its values
// and keeps the other half) followed by plain butterflies -- C-1 + (6 - log2 C) cross-lane moves instead of 6 C.  Afterwards every
// lane holds the total of channel wave_reduce_channel<C>(lane).  Fixed order: bitwise reproducible.
template <int C>
__device__ __forceinline__ int wave_reduce_channel(int lane)
{
    int c = 0;
#pragma unroll
    for (int m = 32, bit = C >> 1; bit >= 1; m >>= 1, bit >>= 1) c |= (lane & m) ? bit : 0;
    return c;
}

template <int C, int HALF, int MASK>
struct WaveHalve {                                             // compile-time recursion: every register index below is static
    static __device__ __forceinline__ void run(float (&acc)[C], int lane)
    {
        const bool up = (lane & MASK) != 0;                    // upper lanes keep the upper half of the channel range
#pragma unroll
        for (int i = 0; i < HALF; i++) {
            const float send = up ? acc[i] : acc[i + HALF], keep = up ? acc[i + HALF] : acc[i];
            acc[i] = keep + __shfl_xor(send, MASK, 64);
        }
        if constexpr (HALF > 1) WaveHalve<C, HALF / 2, MASK / 2>::run(acc, lane);
    }
};

template <int C>
__device__ __forceinline__ float wave_reduce_scatter(float (&acc)[C], int lane)
{
    WaveHalve<C, C / 2, 32>::run(acc, lane);
    float v = acc[0];
#pragma unroll
    for (int m = 32 / C; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Weight gradient of one group of 4 adjacent taps for all C output channels: lanes stride over the rows (samples / symbols), a row's
// 4 input values and C upstream gradients give 4 C MACs (2 C v_pk_fma_f32) per 4 + C LDS reads; then four reduce-scatters.
// out(t, c, sum) is called by one lane per (tap t, channel c).
template <int C, typename OutF>
__device__ __forceinline__ void nn_tapgroup_grad(int n_rows, const float *in, int in_step, const float *g, int g_cstride, bool ones, int lane,
                                                 OutF out)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc[C][2];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c][0] = acc[c][1] = v2f{0.f, 0.f};
    for (int r = lane; r < n_rows; r += 64) {
        const float *ip = in + r * in_step;
        const v2f xA = ones ? v2f{1.f, 0.f} : v2f{ip[0], ip[1]}, xB = ones ? v2f{0.f, 0.f} : v2f{ip[2], ip[3]};
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float gv = g[c * g_cstride + r];
            acc[c][0] += gv * xA;
            acc[c][1] += gv * xB;
        }
    }
    constexpr int WR = 64 / C;
    const int cme = wave_reduce_channel<C>(lane);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float tmp[C];
#pragma unroll
        for (int c = 0; c < C; c++) tmp[c] = (t & 1) ? acc[c][t >> 1].y : acc[c][t >> 1].x;
        const float sum = wave_reduce_scatter<C>(tmp, lane);
        if ((lane & (WR - 1)) == 0) out(t, cme, sum);
    }
}

// the lane-group walks of the two convolutions (mfma_conv16 below; shared by nn_transpose_weights and the callers)
__host__ __device__ inline int conv16_fc1_steps(int k1) { return (((k1 + 1) / 2) + 1) & ~1; }      // taps of a half, rounded up to whole trips
__host__ __device__ inline int conv16_fc2_steps(int k2) { return 4 * k2; }

// ---- transposed weight copies (after every parameter update): w1t[(i k1 + k) C + c], w2t[(cc k2 + k) C + c]
template <int NT, int NLEV>
__device__ __forceinline__ void nn_transpose_weights(const NNLayout &l, int k1, int k2, const float *th, float *w1t, float *w2t,
                                                     float *w2u = nullptr)
{
    constexpr int C = 2 * NLEV;
    if constexpr (nn_mf(NLEV)) {
        // the walk order of mfma_conv16: w1t[(4 t + lg) 16 + c] = fc1.weight[c][lg >> 1][(lg & 1) Th + t] (0 past the half / past k1),
        //                                w2t[(4 t + lg) 16 + c] = fc2.weight[c][4 lg + t / k2][t % k2];  channels / input channels >= C (16-QAM: 8..15): 0
        const int T1 = conv16_fc1_steps(k1), Th = (k1 + 1) / 2;
        for (int j = threadIdx.x; j < 64 * T1; j += NT) {
            const int c = j & 15, lg = (j >> 4) & 3, t = j >> 6, i = lg >> 1, k = (lg & 1) * Th + t;
            w1t[j] = (t < Th && k < k1 && c < C) ? th[l.oW1 + (c * 2 + i) * k1 + k] : 0.f;
        }
        for (int j = threadIdx.x; j < 64 * 4 * k2; j += NT) {
            const int c = j & 15, lg = (j >> 4) & 3, t = j >> 6, cc = 4 * lg + t / k2, k = t % k2;
            w2t[j] = (c < C && cc < C) ? th[l.oW2 + (c * C + cc) * k2 + k] : 0.f;
        }
        if (w2u)
            for (int j = threadIdx.x; j < 256 * k2; j += NT) {
                const int cc = j & 15, c = (j >> 4) & 15, k = j >> 8;
                w2u[j] = (c < C && cc < C) ? th[l.oW2 + (c * C + cc) * k2 + k] : 0.f;
            }
        return;
    } else {
    for (int j = threadIdx.x; j < l.NW1; j += NT) {
        const int c = j % C, ik = j / C, i = ik / k1, k = ik - i * k1;
        w1t[j] = th[l.oW1 + (c * 2 + i) * k1 + k];
    }
    for (int j = threadIdx.x; j < C * C * k2; j += NT) {
        const int c = j % C, r = j / C, cc = r / k2, k = r - cc * k2;
        w2t[j] = th[l.oW2 + (c * C + cc) * k2 + k];
    }
    for (int j = threadIdx.x; j < 7 * C; j += NT) w1t[l.NW1 + j] = w2t[C * C * k2 + j] = 0.f;
    }
    if (w2u)
        for (int j = threadIdx.x; j < C * C * k2; j += NT) {
            const int cc = j % C, r = j / C, c = r % C, k = r / C;
            w2u[j] = th[l.oW2 + (c * C + cc) * k2 + k];
        }
}

// ---- C = 16 (64-QAM): the three convolutions and their weight gradients as GEMMs on v_mfma_f32_16x16x4_f32 -- f32 in, f32 accumulate,
// every output one exact fmaf chain.  Operand maps (cdna_hip_programming.md): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// D[m = 4 (lane >> 4) + reg][n = lane & 15].  M is always the 16 channels.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Sum over the 64 lanes on the vector ALU (DPP row operations + four v_readlane; the form of vaeq_wave.h's wave_sum_dpp): vaeq_common.h's wave_sum is
// a butterfly of six ds_bpermute, i.e. six DEPENDENT LDS round trips -- the BatchNorm statistics take four such sums per channel.  Fixed order.
template <int CTRL>
__device__ __forceinline__ float nn_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_fast(float v)
{
    v += nn_dpp<0xB1>(v);                                      // quad_perm:[1,0,3,2]
    v += nn_dpp<0x4E>(v);                                      // quad_perm:[2,3,0,1]
    v += nn_dpp<0x141>(v);                                     // row_half_mirror
    v += nn_dpp<0x140>(v);                                     // row_mirror: every lane of a 16-lane row holds the row's sum
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// block_reduce3 (vaeq_common.h) with the wave sums above: results in red[0..2] for every thread; red needs 3 (NT / 64) + 4 floats; ends with a barrier
template <int NT>
__device__ __forceinline__ void nn_block_reduce3(float a, float b, float c, float *red)
{
    constexpr int NW = NT / 64;
    a = wave_sum_fast(a);
    b = wave_sum_fast(b);
    c = wave_sum_fast(c);
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

// ldsv: a 4-byte LDS read the compiler must leave where the source puts it (volatile, LDS address space -- as lds2 in vaeq_wave.h).  Used wherever a
// read from a CLAMPED (always valid) address feeds a select: an ordinary load is sunk into a branch of its own behind its own s_waitcnt lgkmcnt(0)
// (the backend will not speculate it), i.e. one exposed LDS round trip per operand -- the pattern round 3 found in the epilogue kernel and, with the ISA
// in hand, here: 35 such branches per tile group of the transposed convolution.
// lds1 = ldsv under -DVAEQ_NN_PIN=1: the MFMA loops below fetch the operands of the NEXT k-steps before the matrix instructions of the current ones, and
// the backend's occupancy-driven scheduler undoes that (each operand read lands directly in front of its v_mfma).  Pinning reads + scheduling fences
// restores the source order (ISA: twelve reads in flight behind ten back-to-back v_mfma) -- and MEASURES 3.7 % SLOWER (212.6 vs 205.0 us per 2048-run
// step): with two waves per SIMD the load-use order of one wave interleaves with the other's matrix passes well enough.  Off by default.
#ifndef VAEQ_NN_PIN
#define VAEQ_NN_PIN 0                                  // measured: the pinned pipeline is 3.7 % SLOWER than the backend's load-use order (see lds1 below)
#endif
typedef const volatile __attribute__((address_space(3))) float lds_cvf;
__device__ __forceinline__ float ldsv(const float *p) { return *(lds_cvf *)p; }
__device__ __forceinline__ float lds1(const float *p)
{
#if VAEQ_NN_PIN
    return *(lds_cvf *)p;
#else
    return *p;
#endif
}
// ... and the matrix instructions must not be hoisted up to "their" loads either (pure operations: the scheduler places each directly behind the
// read that feeds it, which is the same exposed round trip again): nothing crosses this fence
__device__ __forceinline__ void sched_fence()
{
#if VAEQ_NN_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
}
#ifndef VAEQ_NN_PIN_WG
#define VAEQ_NN_PIN_WG 0                               // the same for the weight-gradient loops alone (A/B)
#endif
__device__ __forceinline__ float lds1w(const float *p)
{
#if VAEQ_NN_PIN_WG
    return *(lds_cvf *)p;
#else
    return lds1(p);
#endif
}
__device__ __forceinline__ void sched_fence_w()
{
#if VAEQ_NN_PIN_WG
    __builtin_amdgcn_sched_barrier(0);
#else
    sched_fence();
#endif
}

// Conv1d with 16 output channels:  D[c][col] = bias[c] + sum over the (input row, tap) pairs of  w * in[row rstride + tap + col cstep].
// The 4 k-rows of a v_mfma_f32_16x16x4_f32 (lane group lg = lane >> 4) do NOT take four consecutive (row, tap) pairs: each group WALKS ITS OWN
// sequence -- fc1 (2 input rows): group lg owns row lg >> 1 and the taps of half lg & 1; fc2 (16 rows): group lg owns rows 4 lg .. 4 lg + 3, tap by tap
// -- so that the sample offset of k-step t is  lbase(lane) + off(t)  with off(t) THE SAME for all lanes: a scalar (an instruction immediate once the
// shape is baked) instead of per-lane index arithmetic for every operand read.  On gfx950 the f32 MFMA runs on the vector FMA pipe: a vector
// instruction inside the loop does not hide behind the matrix passes, it adds to them (round 2's loop: 22 vector instructions per 10 MFMAs).
//   off(t) walks:  k = t, t + 1, ... ; at k == kdw: k = 0 and the base advances by rstride   (fc1: kdw = INT_MAX: off(t) = t)
//   wt[(4 t + lg) 16 + c] = weight of channel c for the pair group lg reaches at step t (0 where it has none): nn_transpose_weights
// T k-steps (even: a trip = two k-steps; the operands of the next trip are fetched while the current 2 TB MFMAs run).  Columns past ncols (the last
// tile, and tiles past the last one) are READ -- from padded or neighbouring, always finite LDS cells -- and dropped: no clamps.
// out(c0, col, acc): the lane's channels c0 .. c0 + 3 of column col.
template <int NT, int TB, bool LEAN = false, typename OutF>
__device__ __forceinline__ void mfma_conv16(const float *wt, int T, int lbase, int kdw, int rstride, const float *in, int cstep, int ncols,
                                            const float *bias, OutF out, int creal = 16)
{
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = lane & 15, lg = lane >> 4;
    const int ntile = (ncols + 15) >> 4, K8 = T >> 1;
    // (creal < 16: the channels past it have zero weights and get a zero bias: their outputs are exact zeros)
    const f32x4 b4 = 4 * lg < creal ? *reinterpret_cast<const f32x4 *>(bias + 4 * lg) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tg = wv * TB; tg < ntile; tg += NWV * TB) {
        f32x4 acc[TB];
#pragma unroll
        for (int u = 0; u < TB; u++) acc[u] = b4;
        const float *bp = in + lbase + (tg * 16 + lc) * cstep;          // tile u: + 16 u cstep
        int ob = 0, ok = 0;                                             // off(t) = ob + ok (uniform)
        auto ld = [&](int t, float &a, float (&b)[TB]) {
            a = lds1(wt + 64 * t + lane);
#pragma unroll
            for (int u = 0; u < TB; u++) b[u] = lds1(bp + ob + ok + 16 * u * cstep);
            ok++;
            if (ok == kdw) { ok = 0; ob += rstride; }
        };
        float a0, a1, b0[TB], b1[TB];
        ld(0, a0, b0);
        ld(1, a1, b1);
        auto trip = [&](int t2) {
            const float x0 = a0, x1 = a1;
            float y0[TB], y1[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) { y0[u] = b0[u]; y1[u] = b1[u]; }
            const int tn = 2 * t2 + 2;                                  // past the end: the walk simply continues (finite cells), and the 128 floats
                                                                        // behind wt (another LDS array) are fetched; neither is used
            ld(tn, a0, b0);
            ld(tn + 1, a1, b1);
            sched_fence();
#pragma unroll
            for (int u = 0; u < TB; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, y0[u], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < TB; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, y1[u], acc[u], 0, 0, 0);
            sched_fence();
        };
        if constexpr (LEAN) {
#pragma unroll 1
            for (int t2 = 0; t2 < K8; t2++) trip(t2);
        } else {
            for (int t2 = 0; t2 < K8; t2++) trip(t2);
        }
#pragma unroll
        for (int u = 0; u < TB; u++) {
            const int col = (tg + u) * 16 + lc;
            if (col < ncols) out(4 * lg, col, acc[u]);
        }
    }
}
// Weight gradient of such a convolution:  G[c][j] = sum_{r < nrows} g[c gstride + r] * in[(j / kd) rstride + j % kd + r rstep]  for j < J,
// and the bias gradient G[c][J] = sum_r g[c gstride + r] as one more column.  Waves = (column tile, part of the row range); parts are
// combined through `scratch` (cap floats; every thread of the block must make this call: it may hold a barrier).  Four k-steps of
// operands are fetched per trip, four accumulators take them in turn.
// out(c0, j, acc): the lane's channels c0 .. c0 + 3 of column j <= J.
// ROW1: `in` carries a row of ones as row J / kd (J a multiple of kd) and every operand array may be read up to 16 rows past its end: the bias column is
// then a column like any other, the pointers advance unconditionally and the trip count is a scalar -- per four matrix instructions two vector
// instructions instead of twelve (four selects, a guarded pointer advance, an exec-mask loop).  That matters more than it looks: on gfx950 the f32 MFMA
// runs on the vector FMA pipe, so every vector instruction inside an MFMA loop ADDS to the loop's time instead of hiding behind the matrix passes.
template <int NT, bool ROW1 = false, typename OutF>
__device__ __forceinline__ void mfma_wgrad16(const float *g, int gstride, int nrows, const float *in, int rstep, int J, int kd, int rstride,
                                             float *scratch, int cap, OutF out, int zpad = 0, int onesrow = -1)
{
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = lane & 15, lg = lane >> 4;
    const int ntile = (J + 1 + 15) >> 4, ntw = ntile < NWV ? ntile : NWV;
    int nsplit = NWV / ntw;
    if ((nsplit - 1) * ntw * 256 > cap) nsplit = 1 + cap / (ntw * 256);
    const int tile0 = wv % ntw, part = wv / ntw;
    // ROW1: g is ZERO in the zpad rows behind nrows (zero halo / guard cells of its array): when that covers it, the row range is rounded up to whole trips
    // of four k-steps and no part has a ragged last trip (the clamped path below: ~100 instructions and eight serial reads for two or three k-steps)
    if constexpr (ROW1) {
        const int up = (nrows + 15) & ~15;
        if (up - nrows <= zpad) nrows = up;
    }
    const int steps = (nrows + 3) >> 2, sp = (((steps + nsplit - 1) / nsplit) + 3) & ~3;     // k-steps per part, a multiple of 4
    for (int tile = tile0; tile < ntile; tile += ntw) {            // more than one trip only when ntile > NWV (then nsplit == 1)
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (part < nsplit) {
            const int j = tile * 16 + lc;
            // (ROW1: the bias column J -- and the padding columns behind it, whose results are dropped -- read the row of ones, row `onesrow` of `in`)
            const int jq = j / kd, ji = ROW1 ? (j < J ? jq : (onesrow >= 0 ? onesrow : J / kd)) : (j < J ? jq : 0);
            const int joff = ROW1 ? ji * rstride + (j < J ? j - jq * kd : j - J) : (j < J ? ji * rstride + (j - ji * kd) : 0);
            const bool ones = !ROW1 && j == J;
            const float *gp = g + lc * gstride;
            const int t1 = min(steps, (part + 1) * sp);
            // main trips: four k-steps whose rows all exist -- plain pointer walks, the next trip's operands are fetched while the
            // current four MFMAs run; the (at most one) ragged trip at the end of the row range takes the clamped path below
            const int t0 = part * sp, tfull = nrows >> 2;
            int nmain = max(0, (min(t1, tfull) - t0) >> 2);
            if constexpr (ROW1) nmain = __builtin_amdgcn_readfirstlane(nmain);     // (uniform over the wave: part and tile are)
            const float *pa = gp + 4 * t0 + lg, *pb = in + joff + (4 * t0 + lg) * rstep;
            const int sa = 4, sb = 4 * rstep;
            float an[4], bn[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { an[q] = lds1w(pa + q * sa); bn[q] = lds1w(pb + q * sb); }       // (in range even when nmain == 0: rows < 4 t0 + 16 <= padded arrays)
            if constexpr (ROW1) {
                // two operand sets alternate (no register copies): trip m + 1 is fetched before trip m's matrix instructions, trip m + 2 before those of m + 1
                float a1[4], b1[4];
                auto mma = [&](const float (&x)[4], const float (&y)[4]) {
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[q], acc[q], 0, 0, 0);
                };
                int m = 0;
                for (; m + 2 <= nmain; m += 2) {
                    pa += 4 * sa; pb += 4 * sb;
#pragma unroll
                    for (int q = 0; q < 4; q++) { a1[q] = lds1w(pa + q * sa); b1[q] = lds1w(pb + q * sb); }
                    sched_fence_w();
                    mma(an, bn);
                    sched_fence_w();
                    pa += 4 * sa; pb += 4 * sb;
#pragma unroll
                    for (int q = 0; q < 4; q++) { an[q] = lds1w(pa + q * sa); bn[q] = lds1w(pb + q * sb); }   // (past the last trip: read, never used)
                    sched_fence_w();
                    mma(a1, b1);
                    sched_fence_w();
                }
                if (m < nmain) mma(an, bn);
            } else
            for (int m = 0; m < nmain; m++) {
                float av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { av[q] = an[q]; bv[q] = ones ? 1.0f : bn[q]; }
                if (m + 1 < nmain) { pa += 4 * sa; pb += 4 * sb; }
#pragma unroll
                for (int q = 0; q < 4; q++) { an[q] = lds1w(pa + q * sa); bn[q] = lds1w(pb + q * sb); }
                sched_fence_w();
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[q], 0, 0, 0);
                sched_fence_w();
            }
            for (int t = t0 + 4 * nmain; t < t1; t += 4) {
                float av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {                      // rows past the end (of the array or of this part): clamped reads, zero gradient
                    const int r = 4 * (t + q) + lg, rb = r < nrows ? r : nrows - 1;
                    const float a_ = ldsv(gp + rb), b_ = ldsv(in + joff + rb * rstep);
                    av[q] = (r < nrows && t + q < t1) ? a_ : 0.f;
                    bv[q] = ones ? 1.0f : b_;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[q], 0, 0, 0);
            }
            acc[0] += acc[1]; acc[2] += acc[3]; acc[0] += acc[2];
            if (part > 0) {
                float *sp_ = scratch + ((part - 1) * ntw + tile0) * 256 + lane;
                sp_[0] = acc[0].x; sp_[64] = acc[0].y; sp_[128] = acc[0].z; sp_[192] = acc[0].w;
            }
        }
        if (nsplit > 1) __syncthreads();
        if (part == 0) {
            for (int q = 1; q < nsplit; q++) {
                const float *sp_ = scratch + ((q - 1) * ntw + tile0) * 256 + lane;
                acc[0].x += sp_[0]; acc[0].y += sp_[64]; acc[0].z += sp_[128]; acc[0].w += sp_[192];
            }
            const int j = tile * 16 + lc;
            if (j <= J) out(4 * lg, j, acc[0]);
        }
    }
}

// Backward through the strided Conv1d fc2 (16 -> 16 channels):  gz[cc][s] = sum_{c, k : (s + p2 - k) % sps == 0} w2u[(k 16 + c) 16 + cc] * g2[c GS + (s + p2 - k) / sps],
// one polyphase component of s at a time (for each only every sps-th tap contributes).  g2 = dL/dlogits in the ZERO-GUARDED layout of nn_layout (row
// stride GS, columns -A0 .. 16 ceil(B / 16) + A0 - 1 readable, zeros outside [0, B)): every operand read is unconditional and pinned, the operands of
// the next tap are in flight while the current tap's 4 TB matrix instructions run, and the epilogue reads what it needs of the old buffer (pre) for
// all its outputs before it writes any (post) -- round 2's form had each of these reads in a branch of its own behind its own s_waitcnt lgkmcnt(0)
// (35 exposed LDS round trips per tile group: 4.1 us for 0.8 us of matrix passes).
template <int NT, int TB, typename PreF, typename PostF>
__device__ __forceinline__ void mfma_convT16(const float *w2u, int k2, int p2, int sps, const float *g2, int GS, int L, PreF pre, PostF post)
{
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = lane & 15, lg = lane >> 4;
    for (int ph = 0; ph < sps; ph++) {
        const int kf = (ph + p2) % sps, nk = kf < k2 ? (k2 - kf + sps - 1) / sps : 0;
        const int ncols = (L - ph + sps - 1) / sps, ntile = (ncols + 15) >> 4;      // columns m: s = sps m + ph
        for (int tg = wv * TB; tg < ntile; tg += NWV * TB) {
            f32x4 acc[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *gb = g2 + lg * GS + tg * 16 + lc;                          // + 4 cb GS + 16 u + nsh
            const float *wb = w2u + lg * 16 + lc;                                   // + (k 16 + 4 cb) 16
            float av[2][4], bv[2][4][TB];
            auto ld = [&](int kj, float (&a_)[4], float (&b_)[4][TB]) {
                const int k = kf + kj * sps, nsh = (ph + p2 - k) / sps;             // exact division (may be negative)
#pragma unroll
                for (int cb = 0; cb < 4; cb++) {
                    a_[cb] = lds1(wb + (k * 16 + 4 * cb) * 16);
#pragma unroll
                    for (int u = 0; u < TB; u++) b_[cb][u] = lds1(gb + 4 * cb * GS + 16 * u + nsh);
                }
            };
            auto mm = [&](const float (&a_)[4], const float (&b_)[4][TB]) {
#pragma unroll
                for (int cb = 0; cb < 4; cb++)
#pragma unroll
                    for (int u = 0; u < TB; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[cb], b_[cb][u], acc[u], 0, 0, 0);
            };
            if (nk > 0) ld(0, av[0], bv[0]);
            for (int kj = 0; kj < nk; kj += 2) {                                     // two register sets alternate (no copies)
                if (kj + 1 < nk) ld(kj + 1, av[1], bv[1]);
                sched_fence();
                mm(av[0], bv[0]);
                sched_fence();
                if (kj + 1 < nk) {
                    if (kj + 2 < nk) ld(kj + 2, av[0], bv[0]);
                    sched_fence();
                    mm(av[1], bv[1]);
                    sched_fence();
                }
            }
            float old[TB][4];
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int m = (tg + u) * 16 + lc, mc = m < ncols ? m : ncols - 1;
#pragma unroll
                for (int t = 0; t < 4; t++) old[u][t] = pre(4 * lg + t, sps * mc + ph);
            }
            sched_fence();
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int m = (tg + u) * 16 + lc;
                const float gg[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
                if (m < ncols) {
#pragma unroll
                    for (int t = 0; t < 4; t++) post(4 * lg + t, sps * m + ph, gg[t], old[u][t]);
                }
            }
        }
    }
}

// The same for the samples s in [s_lo, s_lo + s_cnt) only (s_lo a multiple of sps): the half-minibatch kernel below.  out(cc0, s, acc) with the absolute s.
template <int NT, int TB, typename OutF>
__device__ __forceinline__ void mfma_convT16_range(const float *w2u, int k2, int p2, int sps, const float *g2, int B, int s_lo, int s_cnt, OutF out)
{
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = lane & 15, lg = lane >> 4;
    const int m_lo = s_lo / sps;
    for (int ph = 0; ph < sps; ph++) {
        const int kf = (ph + p2) % sps, nk = kf < k2 ? (k2 - kf + sps - 1) / sps : 0;
        const int ncols = (s_cnt - ph + sps - 1) / sps, ntile = (ncols + 15) >> 4;
        for (int tg = wv * TB; tg < ntile; tg += NWV * TB) {
            f32x4 acc[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kj = 0; kj < nk; kj++) {
                const int k = kf + kj * sps, nsh = (ph + p2 - k) / sps;             // exact division (may be negative)
                float av[4], bv[4][TB];
#pragma unroll
                for (int cb = 0; cb < 4; cb++) {
                    const int c = 4 * cb + lg;
                    av[cb] = w2u[(k * 16 + c) * 16 + lc];
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        const int n = m_lo + (tg + u) * 16 + lc + nsh, nc = n < 0 ? 0 : (n < B ? n : B - 1);
                        const float b_ = g2[c * B + nc];
                        bv[cb][u] = (n >= 0 && n < B) ? b_ : 0.f;
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 4; cb++)
#pragma unroll
                    for (int u = 0; u < TB; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb], bv[cb][u], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int m = (tg + u) * 16 + lc;
                if (m < ncols) out(4 * lg, sps * (m_lo + m) + ph, acc[u]);
            }
        }
    }
}

// ---- forward on one LDS-resident window: xs (zero-haloed input) -> z1 (ELU output, zero-haloed) -> a2 (logits).
// item = (channel quad, sample): consecutive lanes take consecutive samples (conflict-free x reads and z1 writes), the four
// channels' weights of a tap come from one 16-byte broadcast read.
template <int NT, int NLEV, bool LEAN = false>
__device__ __forceinline__ void nn_fc1_elu(const NNLayout &l, int k1, const float *xs, const float *th, const float *w1t, float *z1, int Lvalid,
                                           int zlo, int zhi, const float *aff = nullptr)
{
    // z1p[c][p2 + s] for s in [0, Lvalid); entries whose absolute position (zlo + s) lies outside [0, zhi) are fc2's zero padding
    constexpr int C = 2 * NLEV, CQ = C / 4;
    if constexpr (nn_mf(NLEV)) {
        const int lg = (threadIdx.x & 63) >> 4;
        mfma_conv16<NT, (NT >= 1024 ? 2 : 5), LEAN>(w1t, conv16_fc1_steps(k1), (lg >> 1) * l.Lx + (lg & 1) * ((k1 + 1) / 2), 0x7fffffff, 0, xs, 1, Lvalid, th + l.oB1, [&](int c0, int sy, f32x4 acc) {
            const int pos = zlo + sy;
            const bool in = pos >= 0 && pos < zhi;
            const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float z = av[t] > 0.f ? av[t] : __expf(av[t]) - 1.0f;                     // F.elu, alpha = 1 (:177)
                if (aff && c0 + t < C) z = fmaf(aff[c0 + t], z, aff[C + c0 + t]);         // eval-mode BatchNorm: running statistics folded
                z1[(c0 + t) * l.Lz + l.p2 + sy] = in ? z : 0.f;                           // (rows past C: ELU(0) = 0)
            }
        }, C);
        return;
    }
    typedef float v2f __attribute__((ext_vector_type(2)));     // channel pairs: every MAC below is a v_pk_fma_f32
    const int H = (Lvalid + 1) / 2;                            // a thread takes samples sx and sx + H: one weight read feeds 8 MACs
    for (int it = threadIdx.x; it < CQ * H; it += NT) {
        const int cq = it / H, sx = it - cq * H;
        const float4 b = *reinterpret_cast<const float4 *>(th + l.oB1 + 4 * cq);
        v2f a01 = {b.x, b.y}, a23 = {b.z, b.w}, c01 = a01, c23 = a23;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float *xp = xs + i * l.Lx + sx;              // out[s] = sum_k w[k] x[s + k - p1]: the haloed index of x[s + k - p1] is s + k
            const float4 *w = reinterpret_cast<const float4 *>(w1t + (i * k1) * C + 4 * cq);
            for (int k = 0; k < k1; k++) {
                const float xv = xp[k], xw = xp[H + k];
                const float4 w4 = w[k * CQ];
                const v2f w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
                a01 += w01 * xv; a23 += w23 * xv;
                c01 += w01 * xw; c23 += w23 * xw;
            }
        }
        const float av[2][4] = {{a01.x, a01.y, a23.x, a23.y}, {c01.x, c01.y, c23.x, c23.y}};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int sy = sx + u * H;
            if (sy >= Lvalid) continue;
            const int pos = zlo + sy;
            const bool in = pos >= 0 && pos < zhi;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float z = av[u][t] > 0.f ? av[u][t] : __expf(av[u][t]) - 1.0f;           // F.elu, alpha = 1 (:177)
                if (aff) z = fmaf(aff[4 * cq + t], z, aff[C + 4 * cq + t]);              // eval-mode BatchNorm: running statistics folded
                z1[(4 * cq + t) * l.Lz + l.p2 + sy] = in ? z : 0.f;
            }
        }
    }
}

template <int NT, int NLEV, bool LEAN = false>
__device__ __forceinline__ void nn_fc2(const NNLayout &l, int sps, int k2, int Bt, int astride, const float *z1, const float *th, const float *w2t,
                                       float *a2)
{
    constexpr int C = 2 * NLEV, CQ = C / 4;
    if constexpr (nn_mf(NLEV)) {
        const int lg = (threadIdx.x & 63) >> 4;
        mfma_conv16<NT, (NT >= 1024 ? 1 : 3), LEAN>(w2t, conv16_fc2_steps(k2), 4 * lg * l.Lz, k2, l.Lz, z1, sps, Bt, th + l.oB2, [&](int c0, int n, f32x4 acc) {
            a2[(c0 + 0) * astride + n] = acc.x; a2[(c0 + 1) * astride + n] = acc.y;
            a2[(c0 + 2) * astride + n] = acc.z; a2[(c0 + 3) * astride + n] = acc.w;
        }, C);
        return;
    }
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int H = (Bt + 1) / 2;
    for (int it = threadIdx.x; it < CQ * H; it += NT) {
        const int cq = it / H, n = it - cq * H;
        const float4 b = *reinterpret_cast<const float4 *>(th + l.oB2 + 4 * cq);
        v2f a01 = {b.x, b.y}, a23 = {b.z, b.w}, c01 = a01, c23 = a23;
        const float4 *w = reinterpret_cast<const float4 *>(w2t + 4 * cq);
        for (int cc = 0; cc < C; cc++) {
            const float *zp = z1 + cc * l.Lz + n * sps;        // z1[n sps + k - p2] -> haloed index n sps + k
            for (int k = 0; k < k2; k++) {
                const float zv = zp[k], zw = zp[H * sps + k];
                const float4 w4 = w[(cc * k2 + k) * CQ];
                const v2f w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
                a01 += w01 * zv; a23 += w23 * zv;
                c01 += w01 * zw; c23 += w23 * zw;
            }
        }
        a2[(4 * cq + 0) * astride + n] = a01.x; a2[(4 * cq + 1) * astride + n] = a01.y;
        a2[(4 * cq + 2) * astride + n] = a23.x; a2[(4 * cq + 3) * astride + n] = a23.y;
        if (n + H < Bt) {
            a2[(4 * cq + 0) * astride + n + H] = c01.x; a2[(4 * cq + 1) * astride + n + H] = c01.y;
            a2[(4 * cq + 2) * astride + n + H] = c23.x; a2[(4 * cq + 3) * astride + n + H] = c23.y;
        }
    }
}

// SPS = 2 bakes the reference's oversampling factor into the kernel (every / sps and % sps becomes a shift); SPS = 0: run-time sps.
// BK = 1 bakes the sweep script's shape (Eval_run_vaenn.py:25-28: batch_len 300, M = 25, k1 = 25, k2 = 3) into the kernel: the LDS layout and
// every trip count become constants.
template <int NT, int NLEV, bool BN, int SPS, int BK = 0>
// (4-QAM: at most 128 registers, so that two workgroups stay resident per CU -- the kernel sits at that edge)
__global__ __launch_bounds__(NT, (NLEV == 2 && NT <= 512) ? 4 : (NT / 256 > 0 ? NT / 256 : 1)) void nn_train_kernel(const vaeq_nn_args a)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    constexpr int C = 2 * NLEV;
    constexpr bool MF = nn_mf(NLEV);                           // convolutions and their gradients on the 16-row MFMA path
    constexpr int CP = nn_cp(NLEV);                            // channel rows of z1 / zb / a2 in LDS (MF: 16; rows C .. 15 hold zeros)
    const int tid = threadIdx.x, run = blockIdx.x;
    // BK = 2: only the LDS layout of that shape is constant; the shape itself stays in run-time variables (no unrolling on constant trip counts)
    const int B = BK == 1 ? 300 : a.B, sps = SPS ? SPS : a.sps, M = BK == 1 ? 25 : a.M, k1 = BK == 1 ? 25 : a.k1, k2 = BK == 1 ? 3 : a.k2;
    const NNLayout l = BK ? nn_layout(300, SPS ? SPS : 2, 25, NLEV, 25, 3, BN) : nn_layout(B, sps, M, NLEV, k1, k2, BN);
    const int L = l.L, p1 = l.p1, p2 = l.p2, Lx = l.Lx, Lz = l.Lz, mh = l.mh, Mh = l.Mh, nm = l.nm, NP = l.NP;
    float *zb = sm + l.zb, *bnst = sm + l.bnst;                // BN ? separate buffers : zb aliases z1
    float *xs = sm + l.xs, *z1 = sm + l.z1, *a2 = sm + l.a2 + l.A0, *mu = sm + l.mu, *vr = sm + l.vr, *es = sm + l.es, *VS = sm + l.VS;
    const int AS = l.AS;                                       // row stride of a2 (C = 16: zero guard columns around the B logits, nn_layout)
    // residual rows with Mh zeros in front and Mh + 4 behind (row stride ES): the correlations of dL/dh and dL/dmu read e[t] for t in [-Mh, nm + Mh + 2)
    // without clamps, selects or per-lane loop bounds (each of which cost a branch and an exposed LDS round trip per read in round 2's kernel)
    const int ES = l.ES;
    float *esr = es + Mh, *esi = es + ES + Mh;
    float *PH = sm + l.PH;                                     // [M + 1] exclusive prefix sums of |h_j|^2 (the G_V term of dL/dq by two lookups)
    constexpr int UNR_DQ = NLEV == 2 ? 1 : 5;                  // unrolling of dL/dmu's correlation loop (4-QAM: one more register costs a resident workgroup per CU)
    float *th = sm + l.th, *gr = sm + l.gr, *am = sm + l.am, *av = sm + l.av, *ax = sm + l.ax, *w1t = sm + l.w1t, *w2t = sm + l.w2t, *red = sm + l.red;
    float *w2u = sm + l.w2u;
    const float *hs = th + l.oH;                               // h_est[2][M]: re row, im row

    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = a.amp[i];
    const double lr = (double)a.lr[run];
    for (int i = tid; i < NP; i += NT) {
        const size_t g = (size_t)run * NP + i;
        th[i] = a.theta[g]; am[i] = a.adam_m[g]; av[i] = a.adam_v[g]; ax[i] = a.adam_x[g];
    }
    for (int i = tid; i < 2 * Lx; i += NT) xs[i] = 0.f;        // halos stay zero
    for (int i = tid; i < CP * Lz; i += NT) z1[i] = 0.f;
    for (int i = tid; i < CP * AS + 2 * l.A0; i += NT) sm[l.a2 + i] = 0.f;  // guard columns stay zero
    for (int i = tid; i < 2 * ES; i += NT) es[i] = 0.f;
    if (BN) {
        for (int i = tid; i < CP * Lz; i += NT) zb[i] = 0.f;
        for (int i = tid; i < 2 * C; i += NT) bnst[2 * C + i] = a.bn_running[(size_t)run * 2 * C + i];
    }
    if constexpr (MF) {                                        // the rows of ones (nn_layout)
        for (int i = tid; i < Lx; i += NT) xs[2 * Lx + i] = 1.0f;
        for (int i = tid; i < Lz; i += NT) zb[CP * Lz + i] = 1.0f;  // (zb aliases z1 without BatchNorm)
    }
    int step = a.step[run];
    double b1t = pow(0.9, (double)step), b2t = pow(0.999, (double)step);
    __syncthreads();
    nn_transpose_weights<NT, NLEV>(l, k1, k2, th, w1t, w2t, w2u);
    __syncthreads();

    const size_t No = (size_t)a.steps * B;
    const float *rxr = a.rx + (size_t)run * 2 * (size_t)a.S;
    float *qf = a.q_out ? a.q_out + (size_t)run * C * No : nullptr;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NWV = NT / 64, CQ = C / 4;

    // BK = 1: the next minibatch is fetched while the current step computes (its 1200 samples = three per thread wait in registers: xs is read until
    // the last gradient phase) -- a step no longer starts with an exposed HBM round trip
    constexpr int NPRE = BK == 1 ? (1200 + NT - 1) / NT : 1;
    constexpr bool PREF = BK == 1 && !BN && VAEQ_NN_PREF;
    float pre[NPRE];
    auto load_minibatch = [&](int s) {
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int i = tid + u * NT, row = i >= L, c = i - row * L;
            pre[u] = i < 2 * L ? rxr[(size_t)row * a.S + (size_t)s * L + c] : 0.f;
        }
    };
    auto store_minibatch = [&]() {
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int i = tid + u * NT, row = i >= L, c = i - row * L;
            if (i < 2 * L) xs[row * Lx + p1 + c] = pre[u];
        }
    };
    if constexpr (PREF) { load_minibatch(0); store_minibatch(); __syncthreads(); }

#ifdef VAEQ_NN_STAMPS
    __shared__ long long tstamp[16];                            // -DVAEQ_NN_STAMPS: wall-clock stamps of run 0's last step (tools/probe_nn_phases.py)
#define NN_STAMP(i) do { __syncthreads(); if (tid == 0 && run == 0 && s == a.steps - 1) tstamp[i] = wall_clock64(); } while (0)
#else
#define NN_STAMP(i) do { } while (0)
#endif
    for (int s = 0; s < a.steps; s++) {
        NN_STAMP(0);
        // ---- P0: minibatch -> LDS (:276)
        if constexpr (!PREF) {
            for (int i = tid; i < 2 * L; i += NT) {
                const int row = i / L, c = i - row * L;
                xs[row * Lx + p1 + c] = rxr[(size_t)row * a.S + (size_t)s * L + c];
            }
            __syncthreads();
        }
        NN_STAMP(1);
        // ---- P1/P2: fc1 + ELU, fc2
        nn_fc1_elu<NT, NLEV, BK == 1 && VAEQ_NN_LEAN>(l, k1, xs, th, w1t, z1, L, 0, L);
        __syncthreads();
        if (BN) {                                              // BatchNorm1d in training mode (:203): batch statistics over the L samples
            constexpr int NVM = 10;                            // a lane's share of a channel row stays in registers when L <= 640 (the three passes
            const bool inreg = NLEV != 2 && L <= 64 * NVM;     // over the row were three chains of dependent LDS reads; same summation order either way;
                                                               // not for 4-QAM: its 128-register budget -- two workgroups per CU -- has no room for the row)
            for (int c = wv; c < C; c += NWV) {
                float *zr = z1 + c * Lz + p2;
                float zv[NVM];
                float sm_ = 0.f, sv = 0.f;
                if (inreg) {
#pragma unroll
                    for (int u = 0; u < NVM; u++) { const int sx = lane + 64 * u; zv[u] = ldsv(zr + (sx < L ? sx : L - 1)); }
#pragma unroll
                    for (int u = 0; u < NVM; u++) sm_ += lane + 64 * u < L ? zv[u] : 0.f;
                } else
                    for (int sx = lane; sx < L; sx += 64) sm_ += zr[sx];
                const float mean = wave_sum_fast(sm_) / (float)L;
                if (inreg) {
#pragma unroll
                    for (int u = 0; u < NVM; u++) { const float d = zv[u] - mean; sv = lane + 64 * u < L ? fmaf(d, d, sv) : sv; }
                } else
                    for (int sx = lane; sx < L; sx += 64) { const float d = zr[sx] - mean; sv = fmaf(d, d, sv); }
                const float var = wave_sum_fast(sv) / (float)L, rstd = 1.0f / sqrtf(var + 1e-5f);
                if (lane == 0) {
                    bnst[c] = mean; bnst[C + c] = rstd;
                    if (!a.no_update) {                        // running statistics: momentum 0.1, unbiased variance
                        bnst[2 * C + c] = 0.9f * bnst[2 * C + c] + 0.1f * mean;
                        bnst[3 * C + c] = 0.9f * bnst[3 * C + c] + 0.1f * (var * (float)L / (float)(L - 1));
                    }
                }
                const float ga = th[l.oG + c], be = th[l.oBt + c];
                if (inreg) {
#pragma unroll
                    for (int u = 0; u < NVM; u++) {
                        const int sx = lane + 64 * u;
                        const float zh = (zv[u] - mean) * rstd;
                        if (sx < L) { zr[sx] = zh; zb[c * Lz + p2 + sx] = fmaf(ga, zh, be); }
                    }
                } else
                for (int sx = lane; sx < L; sx += 64) {
                    const float zh = (zr[sx] - mean) * rstd;
                    zr[sx] = zh;                               // z1 keeps zhat for the backward pass
                    zb[c * Lz + p2 + sx] = fmaf(ga, zh, be);
                }
            }
            __syncthreads();
        }
        nn_fc2<NT, NLEV, BK == 1 && VAEQ_NN_LEAN>(l, sps, k2, B, AS, zb, th, w2t, a2);
        __syncthreads();
        NN_STAMP(2);
        // ---- P3: per-axis softmax -> q (in place), moments, entropy term; item = (axis, n)
        float klsum = 0.f, vtot = 0.f;
        for (int it = tid; it < 2 * B; it += NT) {
            const int axq = it / B, n = it - axq * B;
            float z[NLEV], zmax = -3.0e38f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] = a2[(axq * NLEV + i) * AS + n]; zmax = fmaxf(zmax, z[i]); }
            float ssum = 0.f;
            float zl[NLEV];                                   // logit - max: log q_i = zl_i - log(ssum) (the reference's + 1e-12 inside the log changes q log(.) by < 1e-12)
#pragma unroll
            for (int i = 0; i < NLEV; i++) { zl[i] = z[i] - zmax; z[i] = __expf(zl[i]); ssum += z[i]; }
            const float rs = 1.0f / ssum, lss = __logf(ssum);
            float e1 = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] *= rs; e1 = fmaf(amp[i], z[i], e1); }
            float e2 = 0.f;
            const bool inr = (n >= mh) && (n < B - mh);        // entropy slice (:90)
#pragma unroll
            for (int i = 0; i < NLEV; i++) {
                const float d = amp[i] - e1;
                e2 = fmaf(z[i] * d, d, e2);
                klsum = inr ? fmaf(z[i], zl[i] - lss, klsum) : klsum;
                a2[(axq * NLEV + i) * AS + n] = z[i];
                if (qf) qf[(size_t)(axq * NLEV + i) * No + (size_t)s * B + n] = z[i];
            }
            mu[it] = e1; vr[it] = e2;
            vtot += e2;
        }
        __syncthreads();
        NN_STAMP(3);
        // ---- P4: residual e = x - D (item t), VS (item j), C
        float se = 0.f;
        if (sps == 2) {
            // D[2 tau + par] = sum_a h[2 a + par] U[tau + mh - a]: the two outputs of a pair share every U read; item = tau
            for (int tau = tid; 2 * tau < nm; tau += NT) {
                float d0r = 0.f, d0i = 0.f, d1r = 0.f, d1i = 0.f;
                const float *ur = mu + tau + mh, *ui = ur + B;
                for (int aa = 0; aa < mh; aa++) {
                    const float a_ = ur[-aa], b_ = ui[-aa];
                    const float c0 = hs[2 * aa], e0 = hs[M + 2 * aa], c1 = hs[2 * aa + 1], e1 = hs[M + 2 * aa + 1];
                    d0r = fmaf(c0, a_, d0r); d0r = fmaf(-e0, b_, d0r);
                    d0i = fmaf(c0, b_, d0i); d0i = fmaf(e0, a_, d0i);
                    d1r = fmaf(c1, a_, d1r); d1r = fmaf(-e1, b_, d1r);
                    d1i = fmaf(c1, b_, d1i); d1i = fmaf(e1, a_, d1i);
                }
                {                                              // a = mh: only the even tap j = Mh exists
                    const float a_ = ur[-mh], b_ = ui[-mh], c0 = hs[Mh], e0 = hs[M + Mh];
                    d0r = fmaf(c0, a_, d0r); d0r = fmaf(-e0, b_, d0r);
                    d0i = fmaf(c0, b_, d0i); d0i = fmaf(e0, a_, d0i);
                }
                const int t = 2 * tau;
                const float er0 = xs[p1 + mh + t] - d0r, ei0 = xs[Lx + p1 + mh + t] - d0i;
                esr[t] = er0; esi[t] = ei0;
                se += er0 * er0 + ei0 * ei0;
                if (t + 1 < nm) {
                    const float er1 = xs[p1 + mh + t + 1] - d1r, ei1 = xs[Lx + p1 + mh + t + 1] - d1i;
                    esr[t + 1] = er1; esi[t + 1] = ei1;
                    se += er1 * er1 + ei1 * ei1;
                }
            }
        } else
        for (int t = tid; t < nm; t += NT) {
            float dr = 0.f, di = 0.f;
            for (int j = (t + Mh) % sps; j <= Mh; j += sps) {
                const int np = (t + Mh - j) / sps;
                const float a_ = mu[np], b_ = mu[B + np], c_ = hs[j], d_ = hs[M + j];
                dr = fmaf(c_, a_, dr); dr = fmaf(-d_, b_, dr);
                di = fmaf(c_, b_, di); di = fmaf(d_, a_, di);
            }
            const float er = xs[p1 + mh + t] - dr, ei = xs[Lx + p1 + mh + t] - di;
            esr[t] = er; esi[t] = ei;
            se += er * er + ei * ei;
        }
        nn_block_reduce3<NT>(se, klsum, vtot, red);           // vtot: this thread's part of sum_n (v_I + v_Q), collected in P3
        float hterm = 0.f;
        // VS[j] = sum over the symbols tap j sees = total - the few it misses at either end.  Those are prefixes of the first / last symbols: wave 0 scans
        // them once (lanes 0-31: symbols 0, 1, ...; lanes 32-63: symbols B - 1, B - 2, ...) and every tap picks its two partial sums with a shuffle -- the
        // per-tap loops of dependent LDS reads (up to 2 x 12 round trips on 25 lanes while seven waves wait at the barrier) cost 1 us of every step
        const int nmiss = (Mh + sps - 1) / sps;                // most symbols a tap misses at one end
        if (nmiss < 32 && nmiss < B) {
            if (tid < 64) {
                const int e = tid & 31, back = tid >> 5, npv = back ? B - 1 - e : e;
                float v = e < nmiss ? ldsv(vr + npv) + ldsv(vr + B + npv) : 0.f;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {             // inclusive scan inside each 32-lane half (fixed order)
                    const float up = __shfl_up(v, d, 32);
                    if (e >= d) v += up;
                }
                const int j = tid < M ? tid : 0, lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps, cnt = B - 1 - hi_;
                const float mf = __shfl(v, lo > 0 ? lo - 1 : 0, 64), mb = __shfl(v, 32 + (cnt > 0 ? cnt - 1 : 0), 64);
                if (tid < M) {
                    VS[j] = red[2] - ((lo > 0 ? mf : 0.f) + (cnt > 0 ? mb : 0.f));
                    hterm = (hs[j] * hs[j] + hs[M + j] * hs[M + j]) * VS[j];
                }
            }
        } else if (tid < M) {
            const int j = tid, lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
            float miss = 0.f;
            for (int np = 0; np < lo; np++) miss += vr[np] + vr[B + np];
            for (int np = hi_ + 1; np < B; np++) miss += vr[np] + vr[B + np];
            VS[j] = red[2] - miss;
            hterm = (hs[j] * hs[j] + hs[M + j] * hs[M + j]) * VS[j];
        }
        if (tid < 64) {                                        // sum_j |h_j|^2 VS[j]: M <= 63 taps, all in wave 0
            float hq = tid < M ? hs[tid] * hs[tid] + hs[M + tid] * hs[M + tid] : 0.f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {                 // inclusive prefix sum over the lanes (fixed order)
                const float up = __shfl_up(hq, d, 64);
                if (tid >= d) hq += up;
            }
            if (tid < M) PH[tid + 1] = hq;
            if (tid == 0) PH[0] = 0.f;
            hterm = wave_sum_fast(hterm);
            if (tid == 0) red[3] = hterm;
        }
        // ---- P5: dL/dh[j] = gC (-2 sum_np e[np sps - Mh + j] conj(mu[np]) + 2 h[j] VS[j]): one wave per group of 4 taps -- a symbol's mu and
        //      4 + 4 adjacent residual samples give 16 FMAs; the 8 partial sums of the group share one reduce-scatter.  The correlation sums need
        //      neither gC nor VS: waves 1 .. NWV - 1 form them WHILE wave 0 runs the scalar section above (it alone used to keep seven waves at
        //      the barrier), leave the raw sums in gr, and 2 M threads scale them once C is known
        constexpr int DHW = NWV > 1 ? NWV - 1 : 1;             // waves that take tap groups
        for (int jg = NWV > 1 ? wv - 1 : 0; jg >= 0 && 4 * jg < M; jg += DHW) {
            const int j0 = 4 * jg, jl = min(j0 + 3, M - 1);
            const int lo = max(0, (Mh - jl + sps - 1) / sps), hi_ = min(B - 1, (nm - 1 + Mh - j0) / sps);
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = 0.f;
            for (int np = lo + lane; np <= hi_; np += 64) {
                const int t0 = np * sps - Mh + j0;
                const float c_ = mu[np], d_ = mu[B + np];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // e[t0 + q]: zero outside [0, nm) by the guard cells -- exactly the symbols tap j0 + q sees (the sums of taps >= M are never read)
                    const float a_ = esr[t0 + q], b_ = esi[t0 + q];
                    acc[2 * q] = fmaf(a_, c_, acc[2 * q]); acc[2 * q] = fmaf(b_, d_, acc[2 * q]);
                    acc[2 * q + 1] = fmaf(b_, c_, acc[2 * q + 1]); acc[2 * q + 1] = fmaf(-a_, d_, acc[2 * q + 1]);
                }
            }
            const float sum = wave_reduce_scatter<8>(acc, lane);
            const int idx = wave_reduce_channel<8>(lane), j = j0 + (idx >> 1), im = idx & 1;
            if ((lane & 7) == 0 && j < M) gr[l.oH + im * M + j] = sum;
        }
        __syncthreads();
        const float Cc = red[0] + red[3];
        const float gC = (float)nm / Cc;
        if (tid == 0 && a.loss) a.loss[(size_t)run * a.steps + s] = (float)nm * logf(Cc) + red[1];
        if (tid < 2 * M) {                                     // (gr is not read before the Adam phase: no barrier needed behind this)
            const int j = tid < M ? tid : tid - M;
            gr[l.oH + tid] = gC * (-2.0f * gr[l.oH + tid] + 2.0f * hs[tid] * VS[j]);
        }
        NN_STAMP(4);
        NN_STAMP(5);
        // ---- P6: dL/dmu, dL/drho -> dL/dq -> softmax backward -> dL/dlogits in place of q; item = n (both axes)
        for (int n = tid; n < B; n += NT) {
            const int sx = n * sps;
            const int jlo = max(0, Mh - sx), jhi = max(jlo - 1, min(Mh, nm - 1 + Mh - sx));
            const float *er = esr + (sx - Mh), *ei = esi + (sx - Mh);
            float pr = 0.f, pi = 0.f;
#pragma unroll UNR_DQ
            for (int j = 0; j <= Mh; j++) {           // (fully unrolled it parks 50 operands in registers: 88 spilled) uniform trip count: terms outside the tap's range meet zero guard cells
                const float a_ = er[j], b_ = ei[j], c_ = hs[j], d_ = hs[M + j];
                pr = fmaf(a_, c_, pr); pr = fmaf(b_, d_, pr);
                pi = fmaf(b_, c_, pi); pi = fmaf(-a_, d_, pi);
            }
            const float gv = gC * (PH[jhi + 1] - PH[jlo]);
            const bool inr = (n >= mh) && (n < B - mh);
#pragma unroll
            for (int axq = 0; axq < 2; axq++) {
                const float gmu = -2.0f * gC * (axq ? pi : pr) - 2.0f * mu[axq * B + n] * gv;   // incl. the -mu^2 part of the variance
                float q[NLEV], gq[NLEV], dot = 0.f;
#pragma unroll
                for (int i = 0; i < NLEV; i++) q[i] = ldsv(a2 + (axq * NLEV + i) * AS + n);   // all reads of the axis first (pinned: the backend otherwise
                                                                                                // issues each in front of its own branch + s_waitcnt lgkmcnt(0))
#pragma unroll
                for (int i = 0; i < NLEV; i++) {
                    gq[i] = amp[i] * gmu + amp[i] * amp[i] * gv;
                    const float ent = __logf(q[i] + 1e-12f) + q[i] * __builtin_amdgcn_rcpf(q[i] + 1e-12f);   // (1-ulp reciprocal: the factor is 1 - 1e-12 / q)
                    gq[i] += inr ? ent : 0.f;                  // a select, not a branch per level
                    dot = fmaf(q[i], gq[i], dot);
                }
#pragma unroll
                for (int i = 0; i < NLEV; i++) a2[(axq * NLEV + i) * AS + n] = q[i] * (gq[i] - dot);
            }
        }
        __syncthreads();
        if constexpr (PREF) { if (s + 1 < a.steps) load_minibatch(s + 1); }
        NN_STAMP(6);
        // ---- P7a: fc2 weight / bias gradients: one wave per (input channel, group of 4 taps); pseudo group at the end: the biases
        if constexpr (MF) {
            // gw2[c][cc][k] = sum_n g2[c][n] zb[cc][n sps + k]: columns j = cc k2 + k, plus the bias column
            mfma_wgrad16<NT, true>(a2, AS, B, zb, sps, C * k2, k2, Lz, mu, 4 * B, [&](int c0, int j, f32x4 acc) {
                const float av_[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (c0 + t >= C) continue;                 // (16-QAM on the 16-row path: rows 8 .. 15 are padding)
                    if (j == C * k2) gr[l.oB2 + c0 + t] = av_[t];
                    else gr[l.oW2 + (c0 + t) * C * k2 + j] = av_[t];
                }
            }, AS - l.A0 - B, CP);
        } else {
            const int nkq = (k2 + 3) / 4, ngrp = C * nkq;
            for (int grp = wv; grp <= ngrp; grp += NWV) {
                const bool bias = grp == ngrp;
                const int cc = bias ? 0 : grp / nkq, k0 = bias ? 0 : (grp - cc * nkq) * 4;
                nn_tapgroup_grad<C>(B, zb + cc * Lz + k0, sps, a2, AS, bias, lane, [&](int t, int c, float sum) {
                    if (bias) { if (t == 0) gr[l.oB2 + c] = sum; }
                    else if (k0 + t < k2) gr[l.oW2 + (c * C + cc) * k2 + k0 + t] = sum;
                });
            }
        }
        __syncthreads();
        NN_STAMP(7);
        // ---- P7b: dL/dz1 through fc2, times ELU' -> dL/da1 in place of z1; item = (4 input channels, sample): every dL/dlogit read
        //      feeds the 4 channels, whose weights come as one 16-byte read of the [k][c][cc] copy
        if constexpr (MF) {
            mfma_convT16<NT, 3>(w2u, k2, p2, sps, a2, AS, L,
                [&](int cc, int sx) { return BN ? 0.f : z1[cc * Lz + p2 + sx]; },
                [&](int cc, int sx, float g, float z) {
                    const int ix = cc * Lz + p2 + sx;
                    if (BN) zb[ix] = g;
                    else z1[ix] = g * (z > 0.f ? 1.0f : z + 1.0f);
                });
        } else
        for (int it = tid; it < CQ * L; it += NT) {
            const int ccq = it / L, sx = it - ccq * L;
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
            for (int k = 0; k < k2; k++) {
                const int t = sx + p2 - k;
                if (t < 0 || t % sps) continue;
                const int n = t / sps;
                if (n >= B) continue;
                const float4 *w = reinterpret_cast<const float4 *>(w2u + (k * C) * C + 4 * ccq);
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float av_ = a2[c * AS + n];
                    const float4 w4 = w[c * CQ];
                    g0 = fmaf(w4.x, av_, g0); g1 = fmaf(w4.y, av_, g1); g2 = fmaf(w4.z, av_, g2); g3 = fmaf(w4.w, av_, g3);
                }
            }
            const float gg[4] = {g0, g1, g2, g3};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ix = (4 * ccq + u) * Lz + p2 + sx;
                if (BN) zb[ix] = gg[u];                                                   // dL/d(BatchNorm output); zb's values are spent
                else {
                    const float z = z1[ix];
                    z1[ix] = gg[u] * (z > 0.f ? 1.0f : z + 1.0f);                         // ELU' = 1 or exp(a1) = z1 + 1
                }
            }
        }
        __syncthreads();
        if (BN) {                                              // BatchNorm backward (batch statistics), then ELU'
            for (int c = wv; c < C; c += NWV) {
                float *zr = z1 + c * Lz + p2;
                const float *gp = zb + c * Lz + p2;
                float s1 = 0.f, s2 = 0.f;
                constexpr int NVM = 10;
                const bool inreg = NLEV != 2 && L <= 64 * NVM; // as in the forward pass: the row's values are read once, into registers
                float gv[NVM], zv[NVM];
                if (inreg) {
#pragma unroll
                    for (int u = 0; u < NVM; u++) {
                        const int sx = lane + 64 * u, sc = sx < L ? sx : L - 1;
                        gv[u] = ldsv(gp + sc); zv[u] = ldsv(zr + sc);
                    }
#pragma unroll
                    for (int u = 0; u < NVM; u++)
                        if (lane + 64 * u < L) { s1 += gv[u]; s2 = fmaf(gv[u], zv[u], s2); }
                } else
                    for (int sx = lane; sx < L; sx += 64) { s1 += gp[sx]; s2 = fmaf(gp[sx], zr[sx], s2); }
                s1 = wave_sum_fast(s1);
                s2 = wave_sum_fast(s2);
                if (lane == 0) { gr[l.oG + c] = s2; gr[l.oBt + c] = s1; }
                const float mean = bnst[c], rstd = bnst[C + c], gs = th[l.oG + c] * rstd, m1 = s1 / (float)L, m2 = s2 / (float)L;
                if (inreg) {
#pragma unroll
                    for (int u = 0; u < NVM; u++) {
                        const int sx = lane + 64 * u;
                        const float zh = zv[u];
                        const float gz = gs * (gv[u] - m1 - zh * m2);
                        const float z = zh / rstd + mean;      // ELU output before the normalisation
                        if (sx < L) zr[sx] = gz * (z > 0.f ? 1.0f : z + 1.0f);
                    }
                } else
                for (int sx = lane; sx < L; sx += 64) {
                    const float zh = zr[sx];
                    const float gz = gs * (gp[sx] - m1 - zh * m2);
                    const float z = zh / rstd + mean;          // ELU output before the normalisation
                    zr[sx] = gz * (z > 0.f ? 1.0f : z + 1.0f);
                }
            }
            __syncthreads();
        }
        NN_STAMP(8);
        // ---- P8: fc1 weight / bias gradients, same scheme: one wave per (input row, group of 4 taps)
        if constexpr (MF) {
            // gw1[c][i][k] = sum_s gz[c][s] x[i][s + k]: columns j = i k1 + k, plus the bias column
            mfma_wgrad16<NT, true>(z1 + p2, Lz, L, xs, 1, 2 * k1, k1, Lx, mu, 4 * B, [&](int c0, int j, f32x4 acc) {
                const float av_[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (c0 + t >= C) continue;
                    if (j == 2 * k1) gr[l.oB1 + c0 + t] = av_[t];
                    else gr[l.oW1 + (c0 + t) * 2 * k1 + j] = av_[t];
                }
            }, Lz - p2 - L);
        } else {
            const int nkq = (k1 + 3) / 4, ngrp = 2 * nkq;
            for (int grp = wv; grp <= ngrp; grp += NWV) {
                const bool bias = grp == ngrp;
                const int i = bias ? 0 : grp / nkq, k0 = bias ? 0 : (grp - i * nkq) * 4;
                // gw1[c][i][k] = sum_s gz[c][s] x[i][s + k - p1] -> haloed index s + k
                nn_tapgroup_grad<C>(L, xs + i * Lx + k0, 1, z1 + p2, Lz, bias, lane, [&](int t, int c, float sum) {
                    if (bias) { if (t == 0) gr[l.oB1 + c] = sum; }
                    else if (k0 + t < k1) gr[l.oW1 + (c * 2 + i) * k1 + k0 + t] = sum;
                });
            }
        }
        __syncthreads();
        NN_STAMP(9);
        // ---- P9: Adam(amsgrad) on every parameter (:285)
        step += 1;
        b1t *= 0.9;
        b2t *= 0.999;
        if constexpr (PREF) { if (s + 1 < a.steps) store_minibatch(); }
        if (!a.no_update) {
            // (hardware reciprocal / square root, 1 ulp each: the step changes by ~2e-7 relative -- as in the DP wave kernel)
            const float rbc2s = (float)(1.0 / sqrt(1.0 - b2t)), ss = (float)(lr / (1.0 - b1t));
            if constexpr (MF) {
                // the owner of a convolution weight also writes it to its places in the transposed copies (the walk order of mfma_conv16, the
                // [k][c][cc] copy of the backward pass): no separate transposition passes, one barrier less; their zero pads are never touched
                const int Th = (k1 + 1) / 2;
                for (int i = tid; i < NP; i += NT) {
                    float w = th[i];
                    adam_update_amsgrad_fast(w, am[i], av[i], ax[i], gr[i], ss, rbc2s);
                    th[i] = w;
                    if (i < l.NW1) {                            // fc1.weight[c][ii][k]
                        const int c = i / (2 * k1), r = i - c * 2 * k1, ii = r / k1, k = r - ii * k1, hh = k >= Th, t = k - hh * Th;
                        w1t[(4 * t + 2 * ii + hh) * 16 + c] = w;
                    } else if (i >= l.oW2 && i < l.oB2) {       // fc2.weight[c][cc][k]
                        const int j = i - l.oW2, c = j / (C * k2), r = j - c * C * k2, cc = r / k2, k = r - cc * k2;
                        w2t[(4 * ((cc & 3) * k2 + k) + (cc >> 2)) * 16 + c] = w;
                        w2u[(k * 16 + c) * 16 + cc] = w;
                    }
                }
            } else {
                for (int i = tid; i < NP; i += NT) adam_update_amsgrad_fast(th[i], am[i], av[i], ax[i], gr[i], ss, rbc2s);
                __syncthreads();
                nn_transpose_weights<NT, NLEV>(l, k1, k2, th, w1t, w2t, w2u);
            }
        }
        __syncthreads();
        NN_STAMP(10);
#ifdef VAEQ_NN_STAMPS
        if (tid < 10 && run == 0 && s == a.steps - 1 && a.loss) a.loss[tid] = (float)(tstamp[tid + 1] - tstamp[tid]);
#endif
    }
    for (int i = tid; i < NP; i += NT) {
        const size_t g = (size_t)run * NP + i;
        if (!a.no_update) { a.theta[g] = th[i]; a.adam_m[g] = am[i]; a.adam_v[g] = av[i]; a.adam_x[g] = ax[i]; }
        if (a.dbg_g) a.dbg_g[g] = gr[i];
    }
    if (tid == 0 && !a.no_update) a.step[run] = step;
    if (BN && !a.no_update)
        for (int i = tid; i < 2 * C; i += NT) a.bn_running[(size_t)run * 2 * C + i] = bnst[2 * C + i];
}

// ---- 64-QAM `Net` (C = 16, sps = 2, B even) with the fc1 activations of HALF a minibatch in LDS: two workgroups per CU.
// nn_train_kernel keeps z1 = ELU(fc1) of the whole minibatch (41 KB) and the three AMSGrad vectors (20 KB) in LDS: 118 KB, one 512-thread
// workgroup per CU, whose eight wavefronts all wait at the same 12 barriers of a step (MFMA pipe 21 % busy, waves waiting 46 % of their cycles).
// Here a workgroup is 256 threads and owns 77 KB: z1 exists for one half of the symbols at a time (forward: half 0, half 1; backward: half 1
// -- still in LDS from the forward pass --, then half 0 after ONE recomputation of fc1 + ELU for it: + 0.24 of the step's 2.1 MMAC, on a kernel
// whose MFMA issue time is a quarter of its step), the AMSGrad vectors stream through registers from / to the caller's arrays once per step
// (40 KB per step and run, loads issued before the last gradient phase), the next minibatch waits in registers.  Two workgroups share a CU,
// each SIMD holds one wave of either: while one waits at a barrier or on LDS the other issues.  Weight-gradient sums run half by half in a
// fixed order (bitwise reproducible; the order differs from nn_train_kernel's, results agree to rounding).
__host__ __device__ inline NNLayout nn_layout_half(int B, int sps, int M, int k1, int k2)
{
    NNLayout l = nn_layout(B, sps, M, 8, k1, k2);
    const int nh = B / 2;
    // z1h[c][j] holds position b0 - p2 + j of the half starting at sample b0 = n0 sps: what fc2 reads for the half's symbols and what ELU' needs
    const int wa = nh * sps, wb = (nh - 1) * sps + l.p2 + 1;
    const int W = l.p2 + (wa > wb ? wa : wb);
    l.Lz = npad4(W + 3);
    while ((l.Lz & 7) != 4) l.Lz += 4;                 // an odd multiple of 4: 16 channel rows x 4 consecutive samples hit 64 different banks
    int o = npad4(l.p2 + 1);                           // front pad: fc1 of half 0 addresses xs[-p2 ...] (masked to fc2's zero padding)
    auto take = [&](int cnt) { int r = o; o += npad4(cnt); return r; };
    l.xs = take(2 * l.Lx);
    l.z1 = take(l.C * l.Lz); l.zb = l.z1; l.bnst = o;
    l.a2 = take(l.C * B);
    l.mu = take(2 * B); l.vr = take(2 * B);
    l.es = take(2 * l.nm);
    l.VS = take(M);
    l.th = take(l.NP);
    l.gr = take(l.NP); l.am = l.av = l.ax = o;
    l.w1t = take(l.NW1 + 7 * l.C);
    l.w2t = take(l.C * l.C * k2 + 7 * l.C);
    l.w2u = take(l.C * l.C * k2);
    l.red = take(64);
    l.total = o;
    return l;
}

template <int NT, int BK>
__global__ __launch_bounds__(NT, 2) void nn_train_half_kernel(const vaeq_nn_args a)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    constexpr int NLEV = 8, C = 16, sps = 2;
    const int tid = threadIdx.x, run = blockIdx.x;
    const int B = BK ? 300 : a.B, M = BK ? 25 : a.M, k1 = BK ? 25 : a.k1, k2 = BK ? 3 : a.k2;
    const NNLayout l = BK ? nn_layout_half(300, 2, 25, 25, 3) : nn_layout_half(B, sps, M, k1, k2);
    const int L = l.L, p1 = l.p1, p2 = l.p2, Lx = l.Lx, Lz = l.Lz, mh = l.mh, Mh = l.Mh, nm = l.nm, NP = l.NP;
    const int nh = B / 2, Lh = nh * sps, W = p2 + max(Lh, Lh - sps + p2 + 1);
    float *xs = sm + l.xs, *z1 = sm + l.z1, *a2 = sm + l.a2, *mu = sm + l.mu, *vr = sm + l.vr, *es = sm + l.es, *VS = sm + l.VS;
    float *th = sm + l.th, *gr = sm + l.gr, *w1t = sm + l.w1t, *w2t = sm + l.w2t, *w2u = sm + l.w2u, *red = sm + l.red;
    const float *hs = th + l.oH;

    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = a.amp[i];
    const double lr = (double)a.lr[run];
    const size_t pbase = (size_t)run * NP;
    for (int i = tid; i < NP; i += NT) th[i] = a.theta[pbase + i];
    for (int i = tid; i < l.xs + 2 * Lx; i += NT) sm[i] = 0.f;  // front pad and halos stay zero
    for (int i = tid; i < C * Lz; i += NT) z1[i] = 0.f;
    int step = a.step[run];
    double b1t = pow(0.9, (double)step), b2t = pow(0.999, (double)step);
    __syncthreads();
    nn_transpose_weights<NT, NLEV>(l, k1, k2, th, w1t, w2t, w2u);

    const size_t No = (size_t)a.steps * B;
    const float *rxr = a.rx + (size_t)run * 2 * (size_t)a.S;
    float *qf = a.q_out ? a.q_out + (size_t)run * C * No : nullptr;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NWV = NT / 64;
    constexpr int NPRE = BK ? (1200 + NT - 1) / NT : 8;        // minibatch samples a thread carries to the next step
    constexpr int NADAM = BK ? (1650 + NT - 1) / NT : 0;       // parameters a thread owns in the Adam phase (run-time shapes: a plain loop)
    float pre[NPRE];
    const bool pre_ok = 2 * L <= NPRE * NT;

    auto load_minibatch = [&](int s) {                         // issued early, parked in registers: xs is read until the last gradient phase
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int i = tid + u * NT, row = i >= L, c = i - row * L;
            pre[u] = i < 2 * L ? rxr[(size_t)row * a.S + (size_t)s * L + c] : 0.f;
        }
    };
    auto store_minibatch = [&]() {
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int i = tid + u * NT, row = i >= L, c = i - row * L;
            if (i < 2 * L) xs[row * Lx + p1 + c] = pre[u];
        }
    };
    auto fc1_half = [&](int b0) {                              // z1h[c][j] = ELU(fc1)(position b0 - p2 + j), zero outside [0, L)
        nn_fc1_elu<NT, NLEV, true>(l, k1, xs + (b0 - p2), th, w1t, z1 - p2, W, b0 - p2, L);
    };
    if (pre_ok) { load_minibatch(0); store_minibatch(); }
    __syncthreads();

#ifdef VAEQ_NN_HALF_STAMPS
    __shared__ long long tstamp[16];
#define NNH_STAMP(i) do { __syncthreads(); if (tid == 0 && run == 0 && s == a.steps - 1) tstamp[i] = wall_clock64(); } while (0)
#else
#define NNH_STAMP(i) do { } while (0)
#endif
    for (int s = 0; s < a.steps; s++) {
        NNH_STAMP(0);
        if (!pre_ok) {
            for (int i = tid; i < 2 * L; i += NT) {
                const int row = i / L, c = i - row * L;
                xs[row * Lx + p1 + c] = rxr[(size_t)row * a.S + (size_t)s * L + c];
            }
            __syncthreads();
        }
        // ---- forward, half by half: fc1 + ELU -> z1h, fc2 -> the half's logits
        for (int hf = 0; hf < 2; hf++) {
            fc1_half(hf * Lh);
            __syncthreads();
            nn_fc2<NT, NLEV, true>(l, sps, k2, nh, B, z1, th, w2t, a2 + hf * nh);
            __syncthreads();
        }
        NNH_STAMP(1);
        if (pre_ok && s + 1 < a.steps) load_minibatch(s + 1);
        // ---- softmax -> q (in place), moments, entropy term; item = (axis, n)
        float klsum = 0.f, vtot = 0.f;
        for (int it = tid; it < 2 * B; it += NT) {
            const int axq = it / B, n = it - axq * B;
            float z[NLEV], zmax = -3.0e38f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] = a2[(axq * NLEV + i) * B + n]; zmax = fmaxf(zmax, z[i]); }
            float ssum = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] = __expf(z[i] - zmax); ssum += z[i]; }
            const float rs = 1.0f / ssum;
            float e1 = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] *= rs; e1 = fmaf(amp[i], z[i], e1); }
            float e2 = 0.f;
            const bool inr = (n >= mh) && (n < B - mh);        // entropy slice (:90)
#pragma unroll
            for (int i = 0; i < NLEV; i++) {
                const float d = amp[i] - e1;
                e2 = fmaf(z[i] * d, d, e2);
                if (inr) klsum = fmaf(z[i], __logf(z[i] + 1e-12f), klsum);
                a2[(axq * NLEV + i) * B + n] = z[i];
                if (qf) qf[(size_t)(axq * NLEV + i) * No + (size_t)s * B + n] = z[i];
            }
            mu[it] = e1; vr[it] = e2;
            vtot += e2;
        }
        __syncthreads();
        NNH_STAMP(2);
        // ---- residual e = x - D (a pair of outputs shares every U read; item = tau), VS, C
        float se = 0.f;
        for (int tau = tid; 2 * tau < nm; tau += NT) {
            float d0r = 0.f, d0i = 0.f, d1r = 0.f, d1i = 0.f;
            const float *ur = mu + tau + mh, *ui = ur + B;
#pragma unroll 1
            for (int aa = 0; aa < mh; aa++) {
                const float a_ = ur[-aa], b_ = ui[-aa];
                const float c0 = hs[2 * aa], e0 = hs[M + 2 * aa], c1 = hs[2 * aa + 1], e1 = hs[M + 2 * aa + 1];
                d0r = fmaf(c0, a_, d0r); d0r = fmaf(-e0, b_, d0r);
                d0i = fmaf(c0, b_, d0i); d0i = fmaf(e0, a_, d0i);
                d1r = fmaf(c1, a_, d1r); d1r = fmaf(-e1, b_, d1r);
                d1i = fmaf(c1, b_, d1i); d1i = fmaf(e1, a_, d1i);
            }
            {
                const float a_ = ur[-mh], b_ = ui[-mh], c0 = hs[Mh], e0 = hs[M + Mh];
                d0r = fmaf(c0, a_, d0r); d0r = fmaf(-e0, b_, d0r);
                d0i = fmaf(c0, b_, d0i); d0i = fmaf(e0, a_, d0i);
            }
            const int t = 2 * tau;
            const float er0 = xs[p1 + mh + t] - d0r, ei0 = xs[Lx + p1 + mh + t] - d0i;
            es[t] = er0; es[nm + t] = ei0;
            se += er0 * er0 + ei0 * ei0;
            if (t + 1 < nm) {
                const float er1 = xs[p1 + mh + t + 1] - d1r, ei1 = xs[Lx + p1 + mh + t + 1] - d1i;
                es[t + 1] = er1; es[nm + t + 1] = ei1;
                se += er1 * er1 + ei1 * ei1;
            }
        }
        block_reduce3<NT>(se, klsum, vtot, red);
        float hterm = 0.f;
        if (tid < M) {
            const int j = tid, lo = (Mh - j + sps - 1) / sps, hi_ = (nm - 1 + Mh - j) / sps;
            float miss = 0.f;
            for (int np = 0; np < lo; np++) miss += vr[np] + vr[B + np];
            for (int np = hi_ + 1; np < B; np++) miss += vr[np] + vr[B + np];
            VS[j] = red[2] - miss;
            hterm = (hs[j] * hs[j] + hs[M + j] * hs[M + j]) * VS[j];
        }
        if (tid < 64) {
            hterm = wave_sum(hterm);
            if (tid == 0) red[3] = hterm;
        }
        __syncthreads();
        const float Cc = red[0] + red[3];
        const float gC = (float)nm / Cc;
        if (tid == 0 && a.loss) a.loss[(size_t)run * a.steps + s] = (float)nm * logf(Cc) + red[1];
        NNH_STAMP(3);
        // ---- dL/dh: one wave per group of 4 taps
        for (int jg = wv; 4 * jg < M; jg += NWV) {
            const int j0 = 4 * jg, jl = min(j0 + 3, M - 1);
            const int lo = max(0, (Mh - jl + sps - 1) / sps), hi_ = min(B - 1, (nm - 1 + Mh - j0) / sps);
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = 0.f;
            for (int np = lo + lane; np <= hi_; np += 64) {
                const int t0 = np * sps - Mh + j0;
                const float c_ = mu[np], d_ = mu[B + np];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int t = t0 + q, tc = t < 0 ? 0 : (t < nm ? t : nm - 1);
                    const bool ok = t >= 0 && t < nm && j0 + q < M;
                    const float ea = ldsv(es + tc), eb = ldsv(es + nm + tc);   // clamped address: read unconditionally, then select
                    const float a_ = ok ? ea : 0.f, b_ = ok ? eb : 0.f;
                    acc[2 * q] = fmaf(a_, c_, acc[2 * q]); acc[2 * q] = fmaf(b_, d_, acc[2 * q]);
                    acc[2 * q + 1] = fmaf(b_, c_, acc[2 * q + 1]); acc[2 * q + 1] = fmaf(-a_, d_, acc[2 * q + 1]);
                }
            }
            const float sum = wave_reduce_scatter<8>(acc, lane);
            const int idx = wave_reduce_channel<8>(lane), j = j0 + (idx >> 1), im = idx & 1;
            if ((lane & 7) == 0 && j < M) gr[l.oH + im * M + j] = gC * (-2.0f * sum + 2.0f * hs[im * M + j] * VS[j]);
        }
        // ---- dL/dmu, dL/drho -> dL/dq -> softmax backward -> dL/dlogits in place of q; item = n (both axes)
        for (int n = tid; n < B; n += NT) {
            const int sx = n * sps;
            const int jlo = max(0, Mh - sx), jhi = min(Mh, nm - 1 + Mh - sx);
            const float *er = es + (sx - Mh), *ei = er + nm;
            float pr = 0.f, pi = 0.f, ph = 0.f;
            for (int j = jlo; j <= jhi; j++) {
                const float a_ = er[j], b_ = ei[j], c_ = hs[j], d_ = hs[M + j];
                pr = fmaf(a_, c_, pr); pr = fmaf(b_, d_, pr);
                pi = fmaf(b_, c_, pi); pi = fmaf(-a_, d_, pi);
                ph = fmaf(c_, c_, ph); ph = fmaf(d_, d_, ph);
            }
            const float gv = gC * ph;
            const bool inr = (n >= mh) && (n < B - mh);
#pragma unroll
            for (int axq = 0; axq < 2; axq++) {
                const float gmu = -2.0f * gC * (axq ? pi : pr) - 2.0f * mu[axq * B + n] * gv;
                float q[NLEV], gq[NLEV], dot = 0.f;
#pragma unroll
                for (int i = 0; i < NLEV; i++) {
                    q[i] = a2[(axq * NLEV + i) * B + n];
                    gq[i] = amp[i] * gmu + amp[i] * amp[i] * gv;
                    if (inr) gq[i] += __logf(q[i] + 1e-12f) + q[i] * __builtin_amdgcn_rcpf(q[i] + 1e-12f);   // (1-ulp reciprocal: the factor is 1 - 1e-12 / q)
                    dot = fmaf(q[i], gq[i], dot);
                }
#pragma unroll
                for (int i = 0; i < NLEV; i++) a2[(axq * NLEV + i) * B + n] = q[i] * (gq[i] - dot);
            }
        }
        __syncthreads();
        NNH_STAMP(4);
        // ---- backward through the network: half 1 (its z1 is still in LDS), then half 0 (fc1 + ELU recomputed)
        float pm[NADAM > 0 ? NADAM : 1], pv[NADAM > 0 ? NADAM : 1], px[NADAM > 0 ? NADAM : 1];
        for (int hb = 1; hb >= 0; hb--) {
            const int b0 = hb * Lh, n0 = hb * nh;
            const bool first = hb == 1;
            if (!first) {
                NNH_STAMP(8);
                fc1_half(b0);
                __syncthreads();
                NNH_STAMP(9);
            }
            // fc2 weight / bias gradients: gw2[c][cc][k] += sum_{n in half} g2[c][n] z1[cc][n sps + k - p2]
            mfma_wgrad16<NT>(a2 + n0, B, nh, z1, sps, C * k2, k2, Lz, mu, 4 * B + 2 * nm, [&](int c0, int j, f32x4 acc) {
                const float av_[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    float *gp = j == C * k2 ? gr + l.oB2 + c0 + t : gr + l.oW2 + (c0 + t) * C * k2 + j;
                    *gp = first ? av_[t] : *gp + av_[t];
                }
            });
            __syncthreads();
            if (first) NNH_STAMP(5); else NNH_STAMP(10);
            // dL/dz1 through fc2, times ELU' -> dL/da1 in place of z1h, for the samples [b0, b0 + Lh)
            mfma_convT16_range<NT, 3>(w2u, k2, p2, sps, a2, B, b0, Lh, [&](int cc0, int sx, f32x4 acc) {
                const float gg[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ix = (cc0 + u) * Lz + p2 + (sx - b0);
                    const float z = z1[ix];
                    z1[ix] = gg[u] * (z > 0.f ? 1.0f : z + 1.0f);
                }
            });
            __syncthreads();
            if (first) NNH_STAMP(6); else NNH_STAMP(11);
            if (!first && NADAM > 0 && !a.no_update) {         // the AMSGrad vectors of this thread's parameters: in flight during the last gradient phase
#pragma unroll
                for (int u = 0; u < NADAM; u++) {
                    const int i = tid + u * NT;
                    const size_t g = pbase + (i < NP ? i : 0);
                    pm[u] = a.adam_m[g]; pv[u] = a.adam_v[g]; px[u] = a.adam_x[g];
                }
            }
            // fc1 weight / bias gradients: gw1[c][i][k] += sum_{s in half} gz[c][s] x[i][s + k - p1]
            mfma_wgrad16<NT>(z1 + p2, Lz, Lh, xs + b0, 1, 2 * k1, k1, Lx, mu, 4 * B + 2 * nm, [&](int c0, int j, f32x4 acc) {
                const float av_[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    float *gp = j == 2 * k1 ? gr + l.oB1 + c0 + t : gr + l.oW1 + (c0 + t) * 2 * k1 + j;
                    *gp = first ? av_[t] : *gp + av_[t];
                }
            });
            __syncthreads();
            if (first) NNH_STAMP(7);
        }
        NNH_STAMP(12);
        // ---- Adam(amsgrad) on every parameter (:285); the next minibatch goes to LDS
        step += 1;
        b1t *= 0.9;
        b2t *= 0.999;
        if (pre_ok && s + 1 < a.steps) store_minibatch();
        if (!a.no_update) {
            const float bc2s = (float)sqrt(1.0 - b2t), ss = (float)(lr / (1.0 - b1t));
            if constexpr (NADAM > 0) {
#pragma unroll
                for (int u = 0; u < NADAM; u++) {
                    const int i = tid + u * NT;
                    if (i < NP) {
                        adam_update_amsgrad(th[i], pm[u], pv[u], px[u], gr[i], ss, bc2s);
                        a.adam_m[pbase + i] = pm[u]; a.adam_v[pbase + i] = pv[u]; a.adam_x[pbase + i] = px[u];
                    }
                }
            } else {
                for (int i = tid; i < NP; i += NT) {
                    float m_ = a.adam_m[pbase + i], v_ = a.adam_v[pbase + i], x_ = a.adam_x[pbase + i];
                    adam_update_amsgrad(th[i], m_, v_, x_, gr[i], ss, bc2s);
                    a.adam_m[pbase + i] = m_; a.adam_v[pbase + i] = v_; a.adam_x[pbase + i] = x_;
                }
            }
            __syncthreads();
            nn_transpose_weights<NT, NLEV>(l, k1, k2, th, w1t, w2t, w2u);
        }
        __syncthreads();
        NNH_STAMP(13);
#ifdef VAEQ_NN_HALF_STAMPS
        if (tid < 13 && run == 0 && s == a.steps - 1 && a.loss) a.loss[tid] = (float)(tstamp[tid + 1] - tstamp[tid]);
#endif
    }
    for (int i = tid; i < NP; i += NT) {
        if (!a.no_update) a.theta[pbase + i] = th[i];
        if (a.dbg_g) a.dbg_g[pbase + i] = gr[i];
    }
    if (tid == 0 && !a.no_update) a.step[run] = step;
}

// ---- eval-mode forward over N symbols in tiles (validation, :293-301): q[R][C][N]
constexpr int NN_TILE = 255;                           // symbols per tile: (255 - 1) sps + k2 = 511 z1 samples at sps = 2, k2 = 3 -> 32 MFMA column tiles = 2 per wave of 1024 threads

template <int NT, int NLEV>
__device__ __forceinline__ void nn_forward_tile(const NNLayout &l, int sps, int k1, int k2, int64_t Ltot, const float *x0, const float *x1,
                                                int n0, int Bt, float *xs, float *z1, float *a2, const float *th, const float *w1t, const float *w2t,
                                                const float *aff = nullptr)
{
    // tile symbols n0 .. n0+Bt-1: z1 needed at absolute positions [n0 sps - p2, (n0+Bt-1) sps + k2 - p2), x p1 beyond that on both sides
    const int tid = threadIdx.x, p1 = l.p1, p2 = l.p2;
    const int zlo = n0 * sps - p2, Lz_need = (Bt - 1) * sps + k2, xlo = zlo - p1, Lx_need = Lz_need + 2 * p1;
    for (int i = tid; i < 2 * (Lx_need + 4); i += NT) {
        const int row = i / (Lx_need + 4), c = i - row * (Lx_need + 4);
        const int64_t sx = (int64_t)xlo + c;
        xs[row * l.Lx + c] = (c < Lx_need && sx >= 0 && sx < Ltot) ? (row ? x1[sx] : x0[sx]) : 0.f;
    }
    __syncthreads();
    // fc1 writes z1p[c][p2 + s] for s in [0, Lz_need) with absolute position zlo + s: shift the base so that p2 + s -> s
    nn_fc1_elu<NT, NLEV>(l, k1, xs, th, w1t, z1 - p2, Lz_need, zlo, (int)Ltot, aff);
    __syncthreads();
    nn_fc2<NT, NLEV>(l, sps, k2, Bt, NN_TILE, z1, th, w2t, a2);   // haloed index of z1[n sps + k - p2] relative to zlo is n sps + k
    __syncthreads();
}

template <int NT, int NLEV>
__global__ __launch_bounds__(NT) void nn_forward_kernel(int N, int sps, int M, int k1, int k2, const float *__restrict__ x,
                                                        const float *__restrict__ theta, const float *__restrict__ bn_running,
                                                        float *__restrict__ q)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    constexpr int C = 2 * NLEV;
    const int tid = threadIdx.x, run = blockIdx.x;
    const bool bn = bn_running != nullptr;
    const NNLayout l = nn_layout(NN_TILE, sps, M, NLEV, k1, k2, bn, true);
    float *xs = sm + l.xs, *z1 = sm + l.z1, *a2 = sm + l.a2, *th = sm + l.th, *w1t = sm + l.w1t, *w2t = sm + l.w2t;
    float *aff = bn ? sm + l.bnst + 4 * C : nullptr;
    for (int i = tid; i < l.NP; i += NT) th[i] = theta[(size_t)run * l.NP + i];
    __syncthreads();
    if (bn && tid < C) {                                       // net.eval(): running statistics folded into one affine map per channel
        const float sc = th[l.oG + tid] / sqrtf(bn_running[(size_t)run * 2 * C + C + tid] + 1e-5f);
        aff[tid] = sc;
        aff[C + tid] = th[l.oBt + tid] - bn_running[(size_t)run * 2 * C + tid] * sc;
    }
    nn_transpose_weights<NT, NLEV>(l, k1, k2, th, w1t, w2t);
    __syncthreads();
    const int64_t Ltot = (int64_t)N * sps;
    const float *x0 = x + (size_t)run * 2 * Ltot, *x1 = x0 + Ltot;
    for (int n0 = 0; n0 < N; n0 += NN_TILE) {
        const int Bt = min(NN_TILE, N - n0);
        nn_forward_tile<NT, NLEV>(l, sps, k1, k2, Ltot, x0, x1, n0, Bt, xs, z1, a2, th, w1t, w2t, aff);
        for (int it = tid; it < 2 * Bt; it += NT) {
            const int axq = it / Bt, n = it - axq * Bt;
            float z[NLEV], zmax = -3.0e38f, ssum = 0.f;
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] = a2[(axq * NLEV + i) * NN_TILE + n]; zmax = fmaxf(zmax, z[i]); }
#pragma unroll
            for (int i = 0; i < NLEV; i++) { z[i] = __expf(z[i] - zmax); ssum += z[i]; }
#pragma unroll
            for (int i = 0; i < NLEV; i++) q[((size_t)run * C + axq * NLEV + i) * N + n0 + n] = z[i] / ssum;
        }
        __syncthreads();
    }
}


// Fused validation pass (:287-301): tiled eval forward, hard decisions = argmax of the logits (first maximum, like argmax q),
// E_q[x_I] of the first 1000 symbols, then the common shift search + SER; q never leaves the chip.
// BK = 1: the sweep script's shape (sps = 2, M = 25, k1 = 25, k2 = 3) baked
template <int NT, int NLEV, int BK = 0>
__global__ __launch_bounds__(NT) void nn_validate_kernel(int N, int sps_, int M_, int k1_, int k2_, int n_shift, const float *__restrict__ x,
                                                         const float *__restrict__ theta, const float *__restrict__ bn_running,
                                                         const float *__restrict__ amp_g, const __half *__restrict__ data,
                                                         float *__restrict__ ser_out, int *__restrict__ shift_out)
{
    extern __shared__ float4 smem4[];
    float *sm = reinterpret_cast<float *>(smem4);
    __shared__ float E[VAL_NE];
    __shared__ float corr[2][VAL_MAXSHIFT];
    __shared__ int sh_s;
    const int tid = threadIdx.x, run = blockIdx.x;
    constexpr int C = 2 * NLEV;
    const int sps = BK ? 2 : sps_, M = BK ? 25 : M_, k1 = BK ? 25 : k1_, k2 = BK ? 3 : k2_;
    const bool bn = bn_running != nullptr;
    const NNLayout l = nn_layout(NN_TILE, sps, M, NLEV, k1, k2, bn, true);
    float *xs = sm + l.xs, *z1 = sm + l.z1, *a2 = sm + l.a2, *th = sm + l.th, *w1t = sm + l.w1t, *w2t = sm + l.w2t, *red = sm + l.red;
    float *aff = bn ? sm + l.bnst + 4 * C : nullptr;
    unsigned char *decs = reinterpret_cast<unsigned char *>(sm + l.total);      // [N] after the forward working set
    float amp[NLEV];
#pragma unroll
    for (int i = 0; i < NLEV; i++) amp[i] = amp_g[i];
    for (int i = tid; i < l.NP; i += NT) th[i] = theta[(size_t)run * l.NP + i];
    __syncthreads();
    if (bn && tid < C) {
        const float sc = th[l.oG + tid] / sqrtf(bn_running[(size_t)run * 2 * C + C + tid] + 1e-5f);
        aff[tid] = sc;
        aff[C + tid] = th[l.oBt + tid] - bn_running[(size_t)run * 2 * C + tid] * sc;
    }
    nn_transpose_weights<NT, NLEV>(l, k1, k2, th, w1t, w2t);
    __syncthreads();
    const int64_t Ltot = (int64_t)N * sps;
    const float *x0 = x + (size_t)run * 2 * Ltot, *x1 = x0 + Ltot;
    const int NE = N < VAL_NE ? N : VAL_NE;
    for (int n0 = 0; n0 < N; n0 += NN_TILE) {
        const int Bt = min(NN_TILE, N - n0);
        nn_forward_tile<NT, NLEV>(l, sps, k1, k2, Ltot, x0, x1, n0, Bt, xs, z1, a2, th, w1t, w2t, aff);
        for (int n = tid; n < Bt; n += NT) {
            int d[2];
#pragma unroll
            for (int axq = 0; axq < 2; axq++) {
                float lg_[NLEV];                               // the axis's logits in one batch of pinned reads (an ordinary load is issued in front of
#pragma unroll                                                 // its own compare behind its own s_waitcnt lgkmcnt(0): eight dependent round trips per axis)
                for (int i = 0; i < NLEV; i++) lg_[i] = ldsv(a2 + (axq * NLEV + i) * NN_TILE + n);
                float best = lg_[0];
                int bi = 0;
#pragma unroll
                for (int i = 1; i < NLEV; i++)
                    if (lg_[i] > best) { best = lg_[i]; bi = i; }
                d[axq] = bi;
                if (axq == 0 && n0 + n < NE) {
                    float ssum = 0.f, e1 = 0.f;
#pragma unroll
                    for (int i = 0; i < NLEV; i++) {
                        const float w = __expf(lg_[i] - best);
                        ssum += w;
                        e1 = fmaf(amp[i], w, e1);
                    }
                    E[n0 + n] = e1 / ssum;
                }
            }
            decs[n0 + n] = (unsigned char)(d[0] | (d[1] << 4));
        }
        __syncthreads();
    }
    validate_tail<NT, NLEV>(N, n_shift, decs, E, NE, data + (size_t)run * 2 * N, red, corr, &sh_s, ser_out + run, shift_out ? shift_out + run : nullptr);
}

template <int NLEV>
static int launch_nn_validate(int R, int N, int sps, int M, int k1, int k2, int n_shift, const float *x, const float *theta, const float *bn,
                              const float *amp, const __half *data, float *ser, int *shift, hipStream_t st)
{
    const size_t lds = (size_t)nn_layout(NN_TILE, sps, M, NLEV, k1, k2, bn != nullptr, true).total * 4 + (((size_t)N + 15) & ~(size_t)15);
    if (lds > 150 * 1024) return VAEQ_ERR_LDS;
    auto k = (NLEV == 8 && sps == 2 && M == 25 && k1 == 25 && k2 == 3) ? nn_validate_kernel<1024, NLEV, NLEV == 8 ? 1 : 0> : nn_validate_kernel<1024, NLEV, 0>;   // (64-QAM only: see launch_nn_train)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    hipLaunchKernelGGL(k, dim3(R), dim3(1024), lds, st, N, sps, M, k1, k2, n_shift, x, theta, bn, amp, data, ser, shift);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

template <int NLEV>
static int launch_nn_train(const vaeq_nn_args &a, size_t lds, hipStream_t st)
{
    void (*k)(const vaeq_nn_args) = a.batch_norm ? nn_train_kernel<512, NLEV, true, 0> : nn_train_kernel<512, NLEV, false, 0>;
    if (a.sps == 2) k = a.batch_norm ? nn_train_kernel<512, NLEV, true, 2> : nn_train_kernel<512, NLEV, false, 2>;
    // the sweep script's shape baked: fully (layout + trip counts) for 64-QAM, +18 % (Net) / +14 % (Net_BN); for 16-QAM only the LDS layout (+10 % / +7 %:
    // with constant trip counts the compiler unrolls into 54-73 spilled registers there, -1 ... -13 %); 4-QAM keeps the run-time shape (either form
    // costs it a resident workgroup per CU: -20 ... -27 %)
    if ((NLEV == 8 || NLEV == 4) && a.sps == 2 && a.B == 300 && a.M == 25 && a.k1 == 25 && a.k2 == 3) {
        constexpr int BK = NLEV == 8 ? 1 : NLEV == 4 ? 2 : 0;
        k = a.batch_norm ? nn_train_kernel<512, NLEV, true, 2, BK> : nn_train_kernel<512, NLEV, false, 2, BK>;
    }
    int nt = 512;
    if constexpr (NLEV == 8) {
        // VAEQ_NN_HALF=1: 64-QAM `Net` at the script's shape on half-minibatch activations, two 256-thread workgroups per CU.  Built, parity-green and
        // MEASURED (profiles/r03/vaenn_half_minibatch_ab.txt): 27.8 vs 28.3 us per run-step at 2048 runs, 48.7 vs 31.9 at <= 256 runs -- not the default.
        const char *he = getenv("VAEQ_NN_HALF");
        if (k == nn_train_kernel<512, NLEV, false, 2, 1> && he && atoi(he) == 1) {
            const size_t ldh = (size_t)nn_layout_half(300, 2, 25, 25, 3).total * 4;
            auto kh = nn_train_half_kernel<256, 1>;
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldh) != hipSuccess)
                return VAEQ_ERR_LDS;
            hipLaunchKernelGGL(kh, dim3(a.R), dim3(256), ldh, st, a);
            return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
        }
        if (const char *e = getenv("VAEQ_NN_NT")) {            // experiment knob: the baked 64-QAM `Net` kernel on 256 / 1024 threads per run
            const int v = atoi(e);
            if ((v == 256 || v == 1024) && k == nn_train_kernel<512, NLEV, false, 2, 1>) {
                nt = v;
                k = v == 256 ? nn_train_kernel<256, NLEV, false, 2, 1> : nn_train_kernel<1024, NLEV, false, 2, 1>;
            }
        }
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    hipLaunchKernelGGL(k, dim3(a.R), dim3(nt), lds, st, a);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

template <int NLEV>
static int launch_nn_forward(int R, int N, int sps, int M, int k1, int k2, const float *x, const float *theta, const float *bn, float *q,
                             hipStream_t st)
{
    const size_t lds = (size_t)nn_layout(NN_TILE, sps, M, NLEV, k1, k2, bn != nullptr, true).total * 4;
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    auto k = nn_forward_kernel<1024, NLEV>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VAEQ_ERR_LDS;
    hipLaunchKernelGGL(k, dim3(R), dim3(1024), lds, st, N, sps, M, k1, k2, x, theta, bn, q);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

static bool nn_shape_ok(int B, int sps, int M, int n_lev, int k1, int k2)
{
    if (B <= 0 || sps <= 0 || sps > 8 || M <= 0 || (M & 1) == 0 || M > 63 || !(n_lev == 2 || n_lev == 4 || n_lev == 8)) return false;
    if (k1 <= 0 || (k1 & 1) == 0 || k1 > 63 || k2 <= 0 || (k2 & 1) == 0 || k2 > 9) return false;
    return (int64_t)B * sps - 2 * (M / 2) > 0 && B > 2 * (M / 2);
}

}  // namespace vaeq

extern "C" int64_t vaeq_nn_param_count(int32_t M, int32_t n_lev, int32_t k1, int32_t k2, int32_t batch_norm)
{
    if (!vaeq::nn_shape_ok(2 * (M / 2) + 1, 1, M, n_lev, k1, k2)) return VAEQ_ERR_SHAPE;
    return vaeq::nn_layout(64, 1, M, n_lev, k1, k2, batch_norm != 0).NP;
}

extern "C" int64_t vaeq_nn_lds_bytes(int32_t B, int32_t sps, int32_t M, int32_t n_lev, int32_t k1, int32_t k2, int32_t batch_norm)
{
    if (!vaeq::nn_shape_ok(B, sps, M, n_lev, k1, k2)) return VAEQ_ERR_SHAPE;
    return (int64_t)vaeq::nn_layout(B, sps, M, n_lev, k1, k2, batch_norm != 0).total * 4;
}

extern "C" int vaeq_nn_train(const vaeq_nn_args *pa, void *stream)
{
    if (!pa) return VAEQ_ERR_NULL;
    const vaeq_nn_args &a = *pa;
    if (a.R == 0) return VAEQ_OK;                              // an empty batch owns no memory: its pointers may be NULL
    if (!a.rx || !a.theta || !a.adam_m || !a.adam_v || !a.adam_x || !a.step || !a.amp || !a.lr) return VAEQ_ERR_NULL;
    if (a.batch_norm && !a.bn_running) return VAEQ_ERR_NULL;
    const int64_t lds = vaeq_nn_lds_bytes(a.B, a.sps, a.M, a.n_lev, a.k1, a.k2, a.batch_norm);
    if (lds < 0) return (int)lds;
    if (lds > 160 * 1024) return VAEQ_ERR_LDS;
    if (a.R < 0 || a.steps <= 0 || (int64_t)a.steps * a.B * a.sps > a.S || (a.batch_norm && (int64_t)a.B * a.sps < 2)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (a.n_lev) {
    case 2: return vaeq::launch_nn_train<2>(a, (size_t)lds, st);
    case 4: return vaeq::launch_nn_train<4>(a, (size_t)lds, st);
    case 8: return vaeq::launch_nn_train<8>(a, (size_t)lds, st);
    }
    return VAEQ_ERR_SHAPE;
}

extern "C" int vaeq_nn_forward(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t k1, int32_t k2, const float *x,
                               const float *theta, const float *bn_running, float *q, void *stream)
{
    if (R == 0 || N == 0) return VAEQ_OK;
    if (!x || !theta || !q) return VAEQ_ERR_NULL;
    if (R < 0 || N < 0 || N > 0x3fffffff || !vaeq::nn_shape_ok(vaeq::NN_TILE, sps, M, n_lev, k1, k2)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (n_lev) {
    case 2: return vaeq::launch_nn_forward<2>(R, (int)N, sps, M, k1, k2, x, theta, bn_running, q, st);
    case 4: return vaeq::launch_nn_forward<4>(R, (int)N, sps, M, k1, k2, x, theta, bn_running, q, st);
    case 8: return vaeq::launch_nn_forward<8>(R, (int)N, sps, M, k1, k2, x, theta, bn_running, q, st);
    }
    return VAEQ_ERR_SHAPE;
}

extern "C" int vaeq_nn_validate(int32_t R, int64_t N, int32_t sps, int32_t M, int32_t n_lev, int32_t k1, int32_t k2, int32_t n_shift,
                                const float *x, const float *theta, const float *bn_running, const float *amp, const void *data_f16, float *ser,
                                int32_t *shift, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!x || !theta || !amp || !data_f16 || !ser) return VAEQ_ERR_NULL;
    if (R < 0 || N < 64 || N > 65536 || n_shift <= 0 || n_shift > vaeq::VAL_MAXSHIFT || !vaeq::nn_shape_ok(vaeq::NN_TILE, sps, M, n_lev, k1, k2))
        return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __half *d = reinterpret_cast<const __half *>(data_f16);
    switch (n_lev) {
    case 2: return vaeq::launch_nn_validate<2>(R, (int)N, sps, M, k1, k2, n_shift, x, theta, bn_running, amp, d, ser, shift, st);
    case 4: return vaeq::launch_nn_validate<4>(R, (int)N, sps, M, k1, k2, n_shift, x, theta, bn_running, amp, d, ser, shift, st);
    case 8: return vaeq::launch_nn_validate<8>(R, (int)N, sps, M, k1, k2, n_shift, x, theta, bn_running, amp, d, ser, shift, st);
    }
    return VAEQ_ERR_SHAPE;
}
