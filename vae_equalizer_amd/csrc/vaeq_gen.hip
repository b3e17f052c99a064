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
#include <hipfft/hipfft.h>
#include <stdint.h>

#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "vaeq.h"
#include "vaeq_common.h"
#include "vaeq_noise.h"
#include "vaeq_wave.h"

namespace vaeq {

// level indices of symbols n (even) and n + 1 of (run, pol): one Philox call, inverse CDF of the PCS pmf on u ~ U(0,1)
__device__ __forceinline__ void draw_symbol_pair(uint64_t seed, uint32_t frame, uint32_t run, int pol, uint32_t n_even, const float *cdf,
                                                 int n_lev, int (&lv)[4])      // (I_n, Q_n, I_n+1, Q_n+1)
{
    const Philox4 r = philox4x32_10(n_even >> 1, run, frame, (uint32_t)(STREAM_SYMBOLS * 2 + pol), (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        lv[c] = 0;
        for (int i = 0; i < n_lev - 1; i++) lv[c] += u[c] >= cdf[i];
    }
}

constexpr int TX_TILE = 2048, TX_NT = 256, TX_MAXG = 96, TX_SYMPH = (TX_TILE / 2 + TX_MAXG / 2 + 16) / 4 + 1, TX_GLO = 24, TX_GHI = 8;

// stage 1 (sps == 2): sig[r][p][s] = sum_n sym[n] g[s + Lg-1 - 2n]   (np.convolve 'valid' of the zero-stuffed symbols with g), s < Ls;
// zeros for Ls <= s < Lrow (row padding for a fast FFT length).  One tile = 2048 samples; the tile's symbols are drawn into LDS
// (4-way polyphase) and -- for the symbols the tile owns -- written out as the TX reference data[r][p][c][n - ref_lo] (fp16).
// Thread t computes the 8 consecutive samples 8t..8t+7: per symbol one LDS read feeds 8 complex MACs; the taps g[kb-6 .. kb+7] of
// four consecutive symbols come from 14 broadcast reads.
// MODE 0: write sig (the DP path: the FFT needs it); with fz.part != NULL also the tile's sum |sig|^2 -> part[run][pol][tile]: the fibre's
//         transfer matrix is unitary at every frequency (shared_funcs.py:38-54: rotations, PMD and CD phases), so the mean power the noise level is
//         derived from after the dispersion (:83) IS the power before it -- no extra pass over the dispersed signal.
//         Single-polarisation path without any sig round trip through HBM:
// MODE 1: only the tile's sum |sig|^2 -> part[run][tile] (fixed-order block reduction);
// MODE 2: sigma from the tile sums (or sigma_fixed), noise added in registers (same Philox words as gen_finish_kernel), planar rx out.
// MODE 3: MODE 1 + the clean samples out as sig[run][s] (complex64, s < Ls) and the TX reference: the frame of vaeq_gen_awgn_clean, whose noise is
//         added where the samples are read (vaeq_awgn_validate_gen) -- symbols drawn and FIR computed once, the noisy frame never written.
struct TxFuse {
    float *part;                 // [R][n_tiles]
    const float *snr_db;         // [R]
    const float *sigma_fixed;    // nullable [R]
    float *rx;                   // [R][2][Lout]
    float *sigma_out;            // nullable [R]
    int Lout, sps;
};

// LDS of one stage-1 workgroup and the two steps every form of the kernel shares: the run's tables, then one tile's symbols + FIR
struct TxShared {
    float2 sym[4 * TX_SYMPH];
    float2 gsp[TX_GLO + TX_MAXG + TX_GHI];
    float cdf[8];
    float amps[8];
};
__device__ __forceinline__ void tx_stage_tables(TxShared &sh, int run, int n_lev, int Lg, const float *__restrict__ amp, const float *__restrict__ cdf_g,
                                                const float2 *__restrict__ g)
{
    const int tid = threadIdx.x;
    if (tid < n_lev) { sh.cdf[tid] = cdf_g[(size_t)run * n_lev + tid]; sh.amps[tid] = amp[tid]; }
    for (int i = tid; i < TX_GLO + TX_MAXG + TX_GHI; i += TX_NT) {
        const int k = i - TX_GLO;
        sh.gsp[i] = (k >= 0 && k < Lg) ? g[k] : make_float2(0.f, 0.f);
    }
    __syncthreads();
}
// tile s0 .. s0 + 2047 of (run, pol): symbols into LDS (and, for the ones the tile owns, out as the TX reference), barrier, then the 8 samples
// 8 tid .. 8 tid + 7 of the tile into acc.  `last`: the row's last tile also owns the symbols past its own half.
__device__ __forceinline__ void tx_tile_fir(TxShared &sh, int s0, bool last, int run, int pol, int npol, int N_conv, int n_lev, int Lg, uint64_t seed,
                                            uint32_t frame, int N, int ref_lo, __half *__restrict__ data, cacc (&acc)[8])
{
    const int tid = threadIdx.x;
    const int nlo = s0 / 2;                                                    // even (TX_TILE / 2 is)
    const int cnt = TX_TILE / 2 + (Lg + 7) / 2 + 8;                            // symbols this tile may touch (<= 4 * TX_SYMPH)
    __half *dI = data ? data + ((size_t)(run * npol + pol) * 2 + 0) * N : nullptr, *dQ = dI ? dI + N : nullptr;
    for (int pi = tid; 2 * pi < cnt; pi += TX_NT) {
        const int n = nlo + 2 * pi;
        int lv[4];
        draw_symbol_pair(seed, frame, run, pol, (uint32_t)n, sh.cdf, n_lev, lv);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int ne = n + e, m = 2 * pi + e;
            const bool in = ne < N_conv;
            const float aI = sh.amps[lv[2 * e]], aQ = sh.amps[lv[2 * e + 1]];
            sh.sym[(m & 3) * TX_SYMPH + (m >> 2)] = in ? make_float2(aI, aQ) : make_float2(0.f, 0.f);
            const int nr = ne - ref_lo;
            if (dI && in && nr >= 0 && nr < N && (m < TX_TILE / 2 || last)) {   // each symbol is owned by exactly one tile
                dI[nr] = __float2half(aI);
                dQ[nr] = __float2half(aQ);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = cacc0();
    const int MW = ((Lg + 7) / 2 + 1 + 3) & ~3;                                // symbols m = 0 .. MW-1 relative to 4 tid
    const float2 *gp = sh.gsp + TX_GLO;
#pragma unroll 1
    for (int m0 = 0; m0 < MW; m0 += 4) {
        const int kb = Lg - 1 - 2 * m0;                                        // sample i of symbol m0 + mm uses tap kb - 2 mm + i
        float2 tp[14];
#pragma unroll
        for (int j = 0; j < 14; j++) tp[j] = gp[kb - 6 + j];
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            const float2 sv = sh.sym[mm * TX_SYMPH + tid + (m0 >> 2)];
#pragma unroll
            for (int i = 0; i < 8; i++) cmac(acc[i], tp[6 - 2 * mm + i].x, tp[6 - 2 * mm + i].y, sv);
        }
    }
}
// AWGN: sample pairs (sb + 2u, sb + 2u + 1) of one tile + noise -> planar rx (noise word j = sample / 2: the same Philox words as gen_finish_kernel;
// the arithmetic of one word lives in awgn_noise_pair, vaeq_noise.h, shared with the validation kernel that adds the noise on load)
__device__ __forceinline__ void tx_noise_store(const cacc (&acc)[8], int sb, float sigma, int run, int pol, uint64_t seed, uint32_t frame, int Lout,
                                               float *__restrict__ rI, float *__restrict__ rQ)
{
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i0 = sb + 2 * u;
        if (i0 >= Lout) break;
        float2 v0 = cfin(acc[2 * u]), v1 = cfin(acc[2 * u + 1]);
        awgn_noise_pair((uint32_t)(i0 >> 1), run, frame, pol, seed, sigma, v0, v1);
        rI[i0] = v0.x;
        rQ[i0] = v0.y;
        if (i0 + 1 < Lout) {
            rI[i0 + 1] = v1.x;
            rQ[i0 + 1] = v1.y;
        }
    }
}

// MODE 3 sends its samples out through the LDS that held the tile's symbols and taps (dead once the FIR is done): 18.4 KB per workgroup instead of 28
union TxSharedOut {
    TxShared sh;
    float2 outs[TX_NT * 9];
};
template <int MODE> struct TxStore { typedef TxShared type; };
template <> struct TxStore<3> { typedef TxSharedOut type; };

template <int MODE>
__global__ __launch_bounds__(TX_NT) void gen_tx_kernel(int N_conv, int n_lev, int Lg, int Ls, int Lrow, const float *__restrict__ amp,
                                                       const float *__restrict__ cdf_g, const float2 *__restrict__ g, uint64_t seed,
                                                       uint32_t frame, int npol, float2 *__restrict__ sig, int N, int ref_lo,
                                                       __half *__restrict__ data, TxFuse fz)
{
    __shared__ typename TxStore<MODE>::type store;
    TxShared &sh = reinterpret_cast<TxShared &>(store);
    const int run = blockIdx.z, pol = blockIdx.y, s0 = blockIdx.x * TX_TILE, tid = threadIdx.x;
    tx_stage_tables(sh, run, n_lev, Lg, amp, cdf_g, g);
    const int sb = s0 + 8 * tid;
    // (MODE 0: every thread stays for the output transpose)
    cacc acc[8];
    tx_tile_fir(sh, s0, s0 + TX_TILE >= Lrow, run, pol, npol, N_conv, n_lev, Lg, seed, frame, N, ref_lo, data, acc);
    __shared__ float red[64];
    if (MODE == 0) {
        // a thread holds 8 CONSECUTIVE samples (64 bytes): written straight out, a wave's store would touch 64 separate 64-byte segments.  They
        // go through LDS (stride 9 per thread: conflict-free both ways) and leave as rows of consecutive 8-byte words.
        __shared__ float2 outs[TX_NT * 9];
        float pw = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float2 v = sb + i < Ls ? cfin(acc[i]) : make_float2(0.f, 0.f);
            outs[9 * tid + i] = v;
            pw += v.x * v.x + v.y * v.y;
        }
        __syncthreads();
        float2 *o = sig + ((size_t)run * npol + pol) * Lrow + s0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = j * TX_NT + tid;
            if (s0 + k < Lrow) o[k] = outs[9 * (k >> 3) + (k & 7)];
        }
        if (fz.part) {                                         // fixed-order block sum: bitwise reproducible
            block_reduce3<TX_NT>(pw, 0.f, 0.f, red);
            if (tid == 0) fz.part[((size_t)run * npol + pol) * gridDim.x + blockIdx.x] = red[0];
        }
        return;
    }
    if (MODE == 1 || MODE == 3) {
        float pw = 0.f;
        float2 *outs3 = reinterpret_cast<float2 *>(&store);   // MODE 3
        if (MODE == 3) __syncthreads();                        // every thread's FIR reads of the symbols are done
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float2 v = cfin(acc[i]);
            if (sb + i < Ls) pw = awgn_power_add(pw, v);
            if (MODE == 3) outs3[9 * tid + i] = v;
        }
        block_reduce3<TX_NT>(pw, 0.f, 0.f, red);              // (its barriers also order the writes of outs3 before the reads below)
        if (tid == 0) fz.part[(size_t)run * gridDim.x + blockIdx.x] = red[0];
        if (MODE == 3) {                                       // clean samples out, through LDS like MODE 0 (rows of consecutive 8-byte words)
            float2 *o = sig + (size_t)run * Lrow + s0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = j * TX_NT + tid;
                if (s0 + k < Ls) o[k] = outs3[9 * (k >> 3) + (k & 7)];
            }
        }
        return;
    }
    float sigma;                                               // MODE 2
    if (fz.sigma_fixed) sigma = fz.sigma_fixed[run];
    else {
        sigma = awgn_sigma_from_parts(fz.part + (size_t)run * gridDim.x, (int)gridDim.x, Ls, fz.sps, fz.snr_db[run]);
    }
    if (fz.sigma_out && blockIdx.x == 0 && tid == 0) fz.sigma_out[run] = sigma;
    float *rI = fz.rx + (size_t)run * 2 * fz.Lout, *rQ = rI + fz.Lout;
    tx_noise_store(acc, sb, sigma, run, pol, seed, frame, fz.Lout, rI, rQ);
}

// AWGN frames of up to NT tiles (training frames: 1200 symbols = 2 tiles) in ONE pass, one workgroup per run: the tiles' clean samples stay in
// registers while their power is summed (per tile, then over the tiles in order: bitwise the sums of the two-pass form), then the noise goes on.
// Against MODE 1 + MODE 2: symbols drawn and FIR computed once instead of twice, one launch.
template <int NT>
__global__ __launch_bounds__(TX_NT) void gen_awgn_onepass_kernel(int N_conv, int n_lev, int Lg, int Ls, const float *__restrict__ amp,
                                                                 const float *__restrict__ cdf_g, const float2 *__restrict__ g, uint64_t seed,
                                                                 uint32_t frame, int N, int ref_lo, __half *__restrict__ data, TxFuse fz)
{
    __shared__ TxShared sh;
    __shared__ float red[64];
    const int run = blockIdx.x, tid = threadIdx.x, ntile = (Ls + TX_TILE - 1) / TX_TILE;
    tx_stage_tables(sh, run, n_lev, Lg, amp, cdf_g, g);
    cacc acc[NT][8];
    float pw = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t < ntile) {                                       // uniform
            tx_tile_fir(sh, t * TX_TILE, t == ntile - 1, run, 0, 1, N_conv, n_lev, Lg, seed, frame, N, ref_lo, data, acc[t]);
            const int sb = t * TX_TILE + 8 * tid;
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float2 v = cfin(acc[t][i]);
                if (sb + i < Ls) p = awgn_power_add(p, v);
            }
            block_reduce3<TX_NT>(p, 0.f, 0.f, red);             // ends with a barrier: the next tile may overwrite the symbols
            pw += red[0];
        }
    }
    const float sigma = fz.sigma_fixed ? fz.sigma_fixed[run] : sqrtf(pw / (float)Ls * (float)fz.sps * 0.5f / exp10f(fz.snr_db[run] * 0.1f));
    if (fz.sigma_out && tid == 0) fz.sigma_out[run] = sigma;
    float *rI = fz.rx + (size_t)run * 2 * fz.Lout, *rQ = rI + fz.Lout;
#pragma unroll
    for (int t = 0; t < NT; t++)
        if (t < ntile) tx_noise_store(acc[t], t * TX_TILE + 8 * tid, sigma, run, 0, seed, frame, fz.Lout, rI, rQ);
}

// any sps: one thread per output sample, symbols and reference drawn per use
__global__ __launch_bounds__(256) void gen_tx_generic_kernel(int N_conv, int sps, int n_lev, int Lg, int Ls, int Lrow, const float *__restrict__ amp,
                                                             const float *__restrict__ cdf_g, const float2 *__restrict__ g, uint64_t seed,
                                                             uint32_t frame, int npol, float2 *__restrict__ sig)
{
    const int run = blockIdx.z, pol = blockIdx.y;
    float cdf[8];
    for (int i = 0; i < n_lev; i++) cdf[i] = cdf_g[(size_t)run * n_lev + i];
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < Lrow; s += gridDim.x * blockDim.x) {
        float ar = 0.f, ai = 0.f;
        const int jhi = s + Lg - 1;                                            // j = s + Lg-1-k must be a multiple of sps
        for (int n = (s + sps - 1) / sps; s < Ls && n * sps <= jhi && n < N_conv; n++) {
            int lv[4];
            draw_symbol_pair(seed, frame, run, pol, (uint32_t)(n & ~1), cdf, n_lev, lv);
            const float2 x = make_float2(amp[lv[2 * (n & 1)]], amp[lv[2 * (n & 1) + 1]]), c = g[jhi - n * sps];
            ar = fmaf(c.x, x.x, ar); ar = fmaf(-c.y, x.y, ar);
            ai = fmaf(c.x, x.y, ai); ai = fmaf(c.y, x.x, ai);
        }
        sig[((size_t)run * npol + pol) * Lrow + s] = make_float2(ar, ai);
    }
}

// TX reference data[r][p][c][n'] = amplitude of symbol n' + lo (shared_funcs.py:89), fp16 -- generic-sps companion of the kernel above
__global__ __launch_bounds__(256) void gen_ref_kernel(int N, int lo, int n_lev, const float *__restrict__ amp, const float *__restrict__ cdf_g,
                                                      uint64_t seed, uint32_t frame, int npol, __half *__restrict__ data)
{
    const int run = blockIdx.z, pol = blockIdx.y;
    float cdf[8];
    for (int i = 0; i < n_lev; i++) cdf[i] = cdf_g[(size_t)run * n_lev + i];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        int lv[4];
        draw_symbol_pair(seed, frame, run, pol, (uint32_t)((n + lo) & ~1), cdf, n_lev, lv);
        const int e = (n + lo) & 1;
        data[((size_t)(run * npol + pol) * 2 + 0) * N + n] = __float2half(amp[lv[2 * e]]);
        data[((size_t)(run * npol + pol) * 2 + 1) * N + n] = __float2half(amp[lv[2 * e + 1]]);
    }
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// stage 2: spectrum of both polarisations times H(f) and the CD phase (shared_funcs.py:40-53), in place
__global__ __launch_bounds__(256) void gen_disperse_kernel(int Ls, double fs_over_Ls, double tau_cd, double tau_pmd, float2 e0, float2 e1,
                                                           float scale, const float *__restrict__ theta, float2 *__restrict__ spec)
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
        const float2 ecd = make_float2(scale * cc, scale * sc), d = make_float2(cd, sd), di = make_float2(cd, -sd);
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

// stage 3a: mean |sig|^2 per run over all polarisations and the first Ls samples of every (Lrow long) row (:83)
__global__ __launch_bounds__(256) void gen_power_kernel(int Ls, int Lrow, int npol, const float2 *__restrict__ sig, float *__restrict__ power)
{
    __shared__ float red[64];
    const int run = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < npol; p++) {
        const float2 *s = sig + ((size_t)run * npol + p) * Lrow;
        for (int i = threadIdx.x; i < Ls; i += 256) acc += s[i].x * s[i].x + s[i].y * s[i].y;
    }
    block_reduce3<256>(acc, 0.f, 0.f, red);
    if (threadIdx.x == 0) power[run] = red[0] / (float)(npol * Ls);
}

// stage 3b: AWGN + planar split: rx[r][p][0/1][s] = Re/Im(sig + sigma_n (n1 + j n2)), s < Lout = sps*N (even)   (:84-88)
// one Philox call -> two Box-Muller pairs -> the noise of two consecutive samples
// stage 1's partial sums of |sig|^2 (all polarisations and tiles of a run, Ls samples per polarisation) -> the run's noise level (:83), in a fixed order;
// the result replaces the run's first partial
__global__ __launch_bounds__(256) void gen_sigma_kernel(int R, int n_parts, int npol, int Ls, int sps, const float *__restrict__ snr_db, float *__restrict__ power)
{
    const int run = blockIdx.x * blockDim.x + threadIdx.x;
    if (run >= R) return;
    float pw = 0.f;
    for (int t = 0; t < n_parts; t++) pw += power[(size_t)run * n_parts + t];
    power[(size_t)run * n_parts] = sqrtf(pw / (float)(npol * Ls) * (float)sps * 0.5f / exp10f(snr_db[run] * 0.1f));
}

// n_parts == 0: power[run] = the mean power (gen_power_kernel); n_parts > 0: power[run * n_parts] = sigma_n (gen_sigma_kernel)
__global__ __launch_bounds__(256) void gen_finish_kernel(int Lrow, int Lout, int sps, const float *__restrict__ snr_db, const float *__restrict__ power,
                                                         uint64_t seed, uint32_t frame, int npol, const float2 *__restrict__ sig,
                                                         float *__restrict__ rx, float *__restrict__ sigma_out,
                                                         const float *__restrict__ sigma_fixed, int n_parts, int Ls)
{
    const int run = blockIdx.z, pol = blockIdx.y;
    // n_parts > 0: gen_sigma_kernel has already turned the run's partial sums into its noise level, left in power[run * n_parts]
    const float sigma = sigma_fixed ? sigma_fixed[run]
                                    : n_parts > 0 ? power[(size_t)run * n_parts] : sqrtf(power[run] * (float)sps * 0.5f / exp10f(snr_db[run] * 0.1f));
    if (sigma_out && pol == 0 && blockIdx.x == 0 && threadIdx.x == 0) sigma_out[run] = sigma;
    const float2 *s = sig + ((size_t)run * npol + pol) * Lrow;
    float *rI = rx + ((size_t)(run * npol + pol) * 2 + 0) * Lout, *rQ = rI + Lout;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; 2 * j < Lout; j += gridDim.x * blockDim.x) {
        const Philox4 r = philox4x32_10((uint32_t)j, run, frame, (uint32_t)(STREAM_NOISE * 2 + pol), (uint32_t)seed, (uint32_t)(seed >> 32));
        float sn0, cs0, sn1, cs1;                               // Box-Muller: two independent N(0,1) per pair of uniforms
        const float rad0 = sigma * sqrtf(-2.0f * __logf(u01(r.x))), rad1 = sigma * sqrtf(-2.0f * __logf(u01(r.z)));
        __sincosf(6.283185307179586f * u01(r.y), &sn0, &cs0);
        __sincosf(6.283185307179586f * u01(r.w), &sn1, &cs1);
        const int i = 2 * j;
        rI[i] = s[i].x + rad0 * cs0;
        rQ[i] = s[i].y + rad0 * sn0;
        if (i + 1 < Lout) {
            rI[i + 1] = s[i + 1].x + rad1 * cs1;
            rQ[i + 1] = s[i + 1].y + rad1 * sn1;
        }
    }
}

// launches of stage 1 / stage 3 shared by the DP and the AWGN entry points
// power_parts (sps == 2 only, nullable): [R][npol][ceil(Lrow / TX_TILE)] partial sums of |sig|^2 written by stage 1
static void launch_tx(int R, int npol, int N, int N_conv, int sps, int n_lev, int Lg, int Ls, int Lrow, int ref_offset, const float *amp,
                      const float *cdf, const float2 *g, uint64_t seed, uint32_t frame, float2 *sig, __half *data, hipStream_t st,
                      float *power_parts = nullptr)
{
    if (sps == 2) {
        TxFuse fz{};
        fz.part = power_parts;
        hipLaunchKernelGGL(gen_tx_kernel<0>, dim3((Lrow + TX_TILE - 1) / TX_TILE, npol, R), dim3(TX_NT), 0, st, N_conv, n_lev, Lg, Ls, Lrow, amp, cdf,
                           g, seed, frame, npol, sig, N, ref_offset, data, fz);
        return;
    }
    hipLaunchKernelGGL(gen_tx_generic_kernel, dim3((Lrow + 255) / 256 > 64 ? 64 : (Lrow + 255) / 256, npol, R), dim3(256), 0, st, N_conv, sps, n_lev,
                       Lg, Ls, Lrow, amp, cdf, g, seed, frame, npol, sig);
    if (data)
        hipLaunchKernelGGL(gen_ref_kernel, dim3((N + 255) / 256 > 64 ? 64 : (N + 255) / 256, npol, R), dim3(256), 0, st, N, ref_offset, n_lev, amp,
                           cdf, seed, frame, npol, data);
}

static void launch_finish(int R, int npol, int N, int sps, int Ls, int Lrow, const float *snr_db, uint64_t seed, uint32_t frame,
                          const float2 *sig, float *power_ws, float *rx, float *sigma_out, hipStream_t st, const float *sigma_fixed = nullptr,
                          int n_parts = 0)
{
    const int Lout = sps * N, nj = (Lout + 1) / 2;
    if (!sigma_fixed && n_parts == 0) hipLaunchKernelGGL(gen_power_kernel, dim3(R), dim3(256), 0, st, Ls, Lrow, npol, sig, power_ws);
    if (!sigma_fixed && n_parts > 0) hipLaunchKernelGGL(gen_sigma_kernel, dim3((R + 255) / 256), dim3(256), 0, st, R, n_parts, npol, Ls, sps, snr_db, power_ws);
    hipLaunchKernelGGL(gen_finish_kernel, dim3((nj + 255) / 256 > 64 ? 64 : (nj + 255) / 256, npol, R), dim3(256), 0, st, Lrow, Lout, sps, snr_db,
                       power_ws, seed, frame, npol, sig, rx, sigma_out, sigma_fixed, n_parts, Ls);
}

}  // namespace vaeq

#include "vaeq_gen_fused.h"

static bool tx_shape_ok(int R, int N, int N_conv, int sps, int n_lev, int Lg, int Ls, int Lrow, int ref_offset)
{
    if (R < 0 || N <= 0 || sps <= 0 || Lg <= 0 || Lg > vaeq::TX_MAXG || !(n_lev == 2 || n_lev == 4 || n_lev == 8) || ref_offset < 0) return false;
    return Ls == sps * (N_conv - 1) + 1 - Lg + 1 && ref_offset + N <= N_conv && Ls >= sps * N && Lrow >= Ls;   // np.convolve 'valid' length
}

extern "C" int vaeq_gen_dp_tx(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t Lrow,
                              int32_t ref_offset, const float *amp, const float *cdf, const float *g_complex, uint64_t seed, uint32_t frame,
                              float *sig_complex, void *data_f16, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!amp || !cdf || !g_complex || !sig_complex) return VAEQ_ERR_NULL;
    if (!tx_shape_ok(R, N, N_conv, sps, n_lev, Lg, Ls, Lrow, ref_offset)) return VAEQ_ERR_SHAPE;
    vaeq::launch_tx(R, 2, N, N_conv, sps, n_lev, Lg, Ls, Lrow, ref_offset, amp, cdf, reinterpret_cast<const float2 *>(g_complex), seed, frame,
                    reinterpret_cast<float2 *>(sig_complex), reinterpret_cast<__half *>(data_f16), reinterpret_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_gen_dp_disperse(int32_t R, int32_t Ls, double fs, double tau_cd, double tau_pmd, float e0_re, float e0_im, float e1_re,
                                    float e1_im, float scale, const float *theta, float *spec_complex, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!theta || !spec_complex) return VAEQ_ERR_NULL;
    if (R < 0 || Ls <= 0) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(vaeq::gen_disperse_kernel, dim3((Ls + 255) / 256, R), dim3(256), 0, st, Ls, fs / (double)Ls, tau_cd, tau_pmd,
                       make_float2(e0_re, e0_im), make_float2(e1_re, e1_im), scale, theta, reinterpret_cast<float2 *>(spec_complex));
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

extern "C" int vaeq_gen_dp_finish(int32_t R, int32_t N, int32_t sps, int32_t Ls, int32_t Lrow, const float *snr_db, uint64_t seed, uint32_t frame,
                                  const float *sig_complex, float *power_ws, float *rx, float *sigma_out, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!snr_db || !sig_complex || !power_ws || !rx) return VAEQ_ERR_NULL;
    if (R < 0 || N <= 0 || sps <= 0 || Ls < sps * N || Lrow < Ls) return VAEQ_ERR_SHAPE;
    vaeq::launch_finish(R, 2, N, sps, Ls, Lrow, snr_db, seed, frame, reinterpret_cast<const float2 *>(sig_complex), power_ws, rx, sigma_out,
                        reinterpret_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

// Single-polarisation AWGN / ISI channel (AWGN_channel/func_VAELE_MQAM_shaping.py:39-61) for R runs: the same three stages with one
// polarisation and no dispersion step: PCS symbols (:45), zero-stuffing + pulse shaping + channel impulse response as one 'valid'
// FIR with g = rrc * h_channel (:47-52), sigma_n from the mean power (:54), complex AWGN (:55), planar rx[R][2][sps*N] (:57) and the
// TX reference data[R][2][N] (fp16, :59).  power_ws: [R][ceil(Ls / 2048)] floats; sig_ws ([R][Ls] complex64) only for sps != 2.
extern "C" int vaeq_gen_awgn(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t ref_offset,
                             const float *amp, const float *cdf, const float *g_complex, const float *snr_db, uint64_t seed, uint32_t frame,
                             float *sig_ws, float *power_ws, float *rx, void *data_f16, float *sigma_out, const float *sigma_fixed,
                             void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!amp || !cdf || !g_complex || (!snr_db && !sigma_fixed) || (sps != 2 && !sig_ws) || !power_ws || !rx) return VAEQ_ERR_NULL;
    if (!tx_shape_ok(R, N, N_conv, sps, n_lev, Lg, Ls, Ls, ref_offset)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float2 *sig = reinterpret_cast<float2 *>(sig_ws);
    if (sps == 2) {                                            // fused: the clean signal never goes through HBM
        const dim3 grid((Ls + vaeq::TX_TILE - 1) / vaeq::TX_TILE, 1, R);
        const float2 *g2 = reinterpret_cast<const float2 *>(g_complex);
        vaeq::TxFuse fz{power_ws, snr_db, sigma_fixed, rx, sigma_out, sps * N, sps};
        const char *two_env = getenv("VAEQ_AWGN_TWOPASS");     // A/B switch: the two-pass form for every frame length
        if (grid.x <= 4 && !(two_env && two_env[0] == '1')) {  // short frames (the training frames of both AWGN scripts): one pass, one workgroup per run
            __half *dh = reinterpret_cast<__half *>(data_f16);
            if (grid.x <= 2)
                hipLaunchKernelGGL(vaeq::gen_awgn_onepass_kernel<2>, dim3(R), dim3(vaeq::TX_NT), 0, st, N_conv, n_lev, Lg, Ls, amp, cdf, g2, seed, frame, N, ref_offset, dh, fz);
            else
                hipLaunchKernelGGL(vaeq::gen_awgn_onepass_kernel<4>, dim3(R), dim3(vaeq::TX_NT), 0, st, N_conv, n_lev, Lg, Ls, amp, cdf, g2, seed, frame, N, ref_offset, dh, fz);
            return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
        }
        if (!sigma_fixed)
            hipLaunchKernelGGL(vaeq::gen_tx_kernel<1>, grid, dim3(vaeq::TX_NT), 0, st, N_conv, n_lev, Lg, Ls, Ls, amp, cdf, g2, seed, frame, 1, sig, N,
                               ref_offset, static_cast<__half *>(nullptr), fz);
        hipLaunchKernelGGL(vaeq::gen_tx_kernel<2>, grid, dim3(vaeq::TX_NT), 0, st, N_conv, n_lev, Lg, Ls, Ls, amp, cdf, g2, seed, frame, 1, sig, N,
                           ref_offset, reinterpret_cast<__half *>(data_f16), fz);
        return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
    }
    vaeq::launch_tx(R, 1, N, N_conv, sps, n_lev, Lg, Ls, Ls, ref_offset, amp, cdf, reinterpret_cast<const float2 *>(g_complex), seed, frame, sig,
                    reinterpret_cast<__half *>(data_f16), st);
    vaeq::launch_finish(R, 1, N, sps, Ls, Ls, snr_db, seed, frame, sig, power_ws, rx, sigma_out, st, sigma_fixed);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

// First stage of vaeq_gen_awgn alone: clean samples, tile power sums, TX reference (include/vaeq.h; consumed by vaeq_awgn_validate_gen)
extern "C" int vaeq_gen_awgn_clean(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t ref_offset,
                                   const float *amp, const float *cdf, const float *g_complex, uint64_t seed, uint32_t frame, float *sig_out,
                                   float *power_ws, void *data_f16, void *stream)
{
    if (R == 0) return VAEQ_OK;
    if (!amp || !cdf || !g_complex || !sig_out || !power_ws) return VAEQ_ERR_NULL;
    if (sps != 2 || !tx_shape_ok(R, N, N_conv, sps, n_lev, Lg, Ls, Ls, ref_offset)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((Ls + vaeq::TX_TILE - 1) / vaeq::TX_TILE, 1, R);
    vaeq::TxFuse fz{};
    fz.part = power_ws;
    hipLaunchKernelGGL(vaeq::gen_tx_kernel<3>, grid, dim3(vaeq::TX_NT), 0, st, N_conv, n_lev, Lg, Ls, Ls, amp, cdf,
                       reinterpret_cast<const float2 *>(g_complex), seed, frame, 1, reinterpret_cast<float2 *>(sig_out), N, ref_offset,
                       reinterpret_cast<__half *>(data_f16), fz);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}

// ---- whole DP frame in one call: stage 1 -> hipFFT (in place) -> stage 2 -> inverse hipFFT (in place, 1/Lrow folded into stage 2)
//      -> stage 3.  Plans are cached per (row length, batch).
namespace vaeq {
static std::mutex g_plan_mu;
static std::map<std::pair<int, int>, hipfftHandle> g_plans;

static int get_plan(int Lrow, int batch, hipfftHandle *out)
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find({Lrow, batch});
    if (it == g_plans.end()) {
        hipfftHandle h;
        int n[1] = {Lrow};
        if (hipfftPlanMany(&h, 1, n, nullptr, 1, Lrow, nullptr, 1, Lrow, HIPFFT_C2C, batch) != HIPFFT_SUCCESS) return VAEQ_ERR_DEVICE;
        it = g_plans.emplace(std::make_pair(Lrow, batch), h).first;
    }
    *out = it->second;
    return VAEQ_OK;
}
}  // namespace vaeq

// ---- the fused three-pass frame (vaeq_gen_fused.h): tables cached per (device, row length, fibre parameters), immutable once built
namespace vaeq {
struct FusedTables { float2 *T; float4 *H; };
struct FusedKey {
    int dev, Lrow;
    double fs, tau_cd, tau_pmd;
    bool operator<(const FusedKey &o) const { return std::tie(dev, Lrow, fs, tau_cd, tau_pmd) < std::tie(o.dev, o.Lrow, o.fs, o.tau_cd, o.tau_pmd); }
};
static std::mutex g_fused_mu;
static std::map<FusedKey, FusedTables> g_fused;

static bool fused_rows(int Lrow) { return Lrow == 4096 || Lrow == 5120 || Lrow == 8192 || Lrow == 10240 || Lrow == 16384 || Lrow == 20480; }

static int get_fused_tables(int Lrow, double fs, double tau_cd, double tau_pmd, hipStream_t st, FusedTables *out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VAEQ_ERR_DEVICE;
    std::lock_guard<std::mutex> lk(g_fused_mu);
    const FusedKey key{dev, Lrow, fs, tau_cd, tau_pmd};
    auto it = g_fused.find(key);
    if (it == g_fused.end()) {
        // entries are never freed: callers launch on the tables after this lock is released (possibly from other threads / on other devices), so
        // there is no point at which an eviction could know them idle; a table pair is 12 bytes x Lrow (240 KB for the default frame) per distinct
        // (device, row length, fibre parameters) -- a sweep over 1000 fibre parameter sets holds 240 MB of the 288 GB
        FusedTables t{nullptr, nullptr};
        if (hipMalloc(&t.T, sizeof(float2) * Lrow) != hipSuccess || hipMalloc(&t.H, sizeof(float4) * Lrow) != hipSuccess) {
            (void)hipFree(t.T);
            return VAEQ_ERR_DEVICE;
        }
        hipLaunchKernelGGL(genf_twiddle_kernel, dim3((Lrow + 255) / 256), dim3(256), 0, st, Lrow, t.T);
        hipLaunchKernelGGL(genf_phase_kernel, dim3((Lrow + 255) / 256), dim3(256), 0, st, Lrow, Lrow / 1024, fs / (double)Lrow, tau_cd, tau_pmd,
                           1.0f / (float)Lrow, t.H);
        if (hipStreamSynchronize(st) != hipSuccess) {          // from here on the tables are read-only: any stream may use them
            (void)hipFree(t.T); (void)hipFree(t.H);
            return VAEQ_ERR_DEVICE;
        }
        it = g_fused.emplace(key, t).first;
    }
    *out = it->second;
    return VAEQ_OK;
}

template <int N1>
static void launch_fused(int R, int N, int N_conv, int n_lev, int Lg, int Ls, int ref_offset, const float *amp, const float *cdf, const float2 *g,
                         const float *snr_db, const float *theta, float2 E00, float2 E01, float2 E11, uint64_t seed, uint32_t frame, float2 *sig,
                         float *power_ws, float *rx, __half *data, float *sigma_out, const FusedTables &tb, hipStream_t st)
{
    hipLaunchKernelGGL(genf_tx_kernel<N1>, dim3(4, 2, R), dim3(GF_NT), 0, st, N_conv, n_lev, Lg, Ls, amp, cdf, g, seed, frame, sig, N, ref_offset, data,
                       power_ws);
    hipLaunchKernelGGL(gen_sigma_kernel, dim3((R + 255) / 256), dim3(256), 0, st, R, GF_PARTS, 2, Ls, 2, snr_db, power_ws);
    const int rpw = R >= 4096 ? 4 : R >= 1024 ? 2 : 1;       // runs per wavefront: amortises the wave's twiddle set-up once the chip is full
    hipLaunchKernelGGL(genf_fft_kernel<N1>, dim3((R + 4 * rpw - 1) / (4 * rpw), N1), dim3(256), 0, st, R, rpw, tb.T, tb.H, E00, E01, E11, theta, sig);
    hipLaunchKernelGGL(genf_finish_kernel<N1>, dim3(4, 2, R), dim3(GF_NT), 0, st, 2 * N, power_ws, GF_PARTS, seed, frame, sig, rx, sigma_out);
}
}  // namespace vaeq

extern "C" int32_t vaeq_gen_dp_power_parts(int32_t Lrow)
{
    if (Lrow <= 0) return VAEQ_ERR_SHAPE;
    const int staged = 2 * ((Lrow + vaeq::TX_TILE - 1) / vaeq::TX_TILE);
    return staged > vaeq::GF_PARTS ? staged : vaeq::GF_PARTS;
}

extern "C" int vaeq_gen_dp_frame(int32_t R, int32_t N, int32_t N_conv, int32_t sps, int32_t n_lev, int32_t Lg, int32_t Ls, int32_t Lrow,
                                 int32_t ref_offset, const float *amp, const float *cdf, const float *g_complex, const float *snr_db,
                                 const float *theta, double fs, double tau_cd, double tau_pmd, float e0_re, float e0_im, float e1_re,
                                 float e1_im, uint64_t seed, uint32_t frame, float *sig_ws, float *power_ws, float *rx, void *data_f16,
                                 float *sigma_out, void *stream)
{
    if (R == 0) return VAEQ_OK;                                // an empty batch owns no memory: its pointers may be NULL
    if (!amp || !cdf || !g_complex || !snr_db || !theta || !sig_ws || !power_ws || !rx) return VAEQ_ERR_NULL;
    if (!tx_shape_ok(R, N, N_conv, sps, n_lev, Lg, Ls, Lrow, ref_offset)) return VAEQ_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float2 *sig = reinterpret_cast<float2 *>(sig_ws);
    const char *staged_env = getenv("VAEQ_GEN_STAGED");      // A/B switch: the five-pass hipFFT chain for every row length
    if (sps == 2 && vaeq::fused_rows(Lrow) && !(staged_env && staged_env[0] == '1')) {
        vaeq::FusedTables tb;
        const int rc = vaeq::get_fused_tables(Lrow, fs, tau_cd, tau_pmd, st, &tb);
        if (rc != VAEQ_OK) return rc;
        // E00 = e0^2, E01 = e0 e1, E11 = e1^2: with u = c a + s b, v = c b - s a (c, s = cos, sin theta) and d = e^{j pi f tau_pmd} the fibre matrix
        // R^T diag(d, d*) R of shared_funcs.py:47-52 applied to (a, b) is (c E00 d u - s E01 d* v,  s E01 d u + c E11 d* v)
        const float2 e0 = make_float2(e0_re, e0_im), e1 = make_float2(e1_re, e1_im);
        auto hmul = [](float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); };
        const float2 E00 = hmul(e0, e0), E01 = hmul(e0, e1), E11 = hmul(e1, e1);
        const float2 *g2 = reinterpret_cast<const float2 *>(g_complex);
        __half *data = reinterpret_cast<__half *>(data_f16);
#define VAEQ_FUSED(N1) vaeq::launch_fused<N1>(R, N, N_conv, n_lev, Lg, Ls, ref_offset, amp, cdf, g2, snr_db, theta, E00, E01, E11, seed, frame, sig, power_ws, rx, data, sigma_out, tb, st)
        switch (Lrow / 1024) {
        case 4: VAEQ_FUSED(4); break;
        case 5: VAEQ_FUSED(5); break;
        case 8: VAEQ_FUSED(8); break;
        case 10: VAEQ_FUSED(10); break;
        case 16: VAEQ_FUSED(16); break;
        default: VAEQ_FUSED(20); break;
        }
#undef VAEQ_FUSED
        return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
    }
    hipfftHandle plan;
    const int rc = vaeq::get_plan(Lrow, 2 * R, &plan);
    if (rc != VAEQ_OK) return rc;
    if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS) return VAEQ_ERR_DEVICE;
    // sps == 2: stage 1 also leaves the tiles' sums of |sig|^2 in power_ws (the dispersion is unitary: no power pass after it)
    const int n_parts = sps == 2 ? 2 * ((Lrow + vaeq::TX_TILE - 1) / vaeq::TX_TILE) : 0;
    vaeq::launch_tx(R, 2, N, N_conv, sps, n_lev, Lg, Ls, Lrow, ref_offset, amp, cdf, reinterpret_cast<const float2 *>(g_complex), seed, frame, sig,
                    reinterpret_cast<__half *>(data_f16), st, n_parts ? power_ws : nullptr);
    if (hipfftExecC2C(plan, reinterpret_cast<hipfftComplex *>(sig), reinterpret_cast<hipfftComplex *>(sig), HIPFFT_FORWARD) != HIPFFT_SUCCESS)
        return VAEQ_ERR_LAUNCH;
    hipLaunchKernelGGL(vaeq::gen_disperse_kernel, dim3((Lrow + 255) / 256, R), dim3(256), 0, st, Lrow, fs / (double)Lrow, tau_cd, tau_pmd,
                       make_float2(e0_re, e0_im), make_float2(e1_re, e1_im), 1.0f / (float)Lrow, theta, sig);
    if (hipfftExecC2C(plan, reinterpret_cast<hipfftComplex *>(sig), reinterpret_cast<hipfftComplex *>(sig), HIPFFT_BACKWARD) != HIPFFT_SUCCESS)
        return VAEQ_ERR_LAUNCH;
    vaeq::launch_finish(R, 2, N, sps, Ls, Lrow, snr_db, seed, frame, sig, power_ws, rx, sigma_out, st, nullptr, n_parts);
    return hipGetLastError() == hipSuccess ? VAEQ_OK : VAEQ_ERR_LAUNCH;
}
