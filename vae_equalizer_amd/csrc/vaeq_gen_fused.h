// vaeq_gen_fused.h -- the DP channel simulator's frame in THREE passes over the signal instead of five, for row lengths
// Lrow = N1 * 1024 (N1 in {4, 5, 8, 10, 16, 20}; the default frame is 20 * 1024).  Included by vaeq_gen.hip (Philox, symbol draw, constants).
//
// Same model as the staged path (vaeq_gen_dp_tx -> FFT -> vaeq_gen_dp_disperse -> inverse FFT -> vaeq_gen_dp_finish, i.e.
// optical_DP_channel/shared_funcs.py:65-90 with :38-54 applied in the frequency domain), same Philox words for symbols and noise; the Lrow-point
// transform is split as N1 x 1024 (Cooley-Tukey, decimation in time on the way in, in frequency on the way back) so that its two outer
// stages need no transposition and ride on the producer and the consumer of the signal:
//
//   pass A  genf_tx_kernel      thread n2 in [0, 1024) computes the pulse-shaped samples x[1024 n1 + n2], n1 < N1, in registers (symbols of
//                               the N1 stripes staged in LDS, its polyphase taps read once per tap for all N1 samples), takes the N1-point
//                               DFT over n1 in registers and writes Y[k1][n2]                                    (write 8 B / sample)
//   pass B  genf_fft_kernel     one wavefront per (run, k1): rows k1 of both polarisations (1024 points, 16 per lane), twiddle
//                               W_L^{k1 n2}, forward 1024-point FFT (radix 16 x 16 x 4, two exchanges through the wave's LDS slice, no
//                               barrier), the 2x2 fibre matrix at the frequencies k1 + N1 k2 (per-frequency phases from a table built once
//                               per parameter set), inverse 1024-point FFT, twiddle W_L^{-k1 m1}, in place          (read + write 8 B)
//   pass C  genf_finish_kernel  thread m1 reads Z[k1][m1], k1 < N1, takes the inverse N1-point DFT in registers = samples x[m1 + 1024 m2],
//                               adds the noise and writes the planar rx                                          (read 8 B, write 8 B)
//
// 13 GB instead of 27 GB through HBM per 8192-run frame, no hipFFT plan (whose creation costs seconds on a process's first frame).
#pragma once

namespace vaeq {

// ---- small DFTs on register arrays (forward: e^{-2 pi i nk/N}; INV: conjugate kernel, no 1/N) -----------------------------------------------
// complex products as TWO packed instructions, sums with +-j b as ONE: the half swaps and signs ride on op_sel / neg_lo / neg_hi (the backend
// does not find these forms: it swaps with v_mov and multiplies the halves as scalars)
__device__ __forceinline__ v2f cmulv(v2f a, v2f b)                             // a * b
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                                            // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t)); // + (-a.y b.y, a.y b.x)
    return r;
}
__device__ __forceinline__ v2f cmulc(v2f a, v2f b)                             // a * conj(b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                               // (a.x b.x, -a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));               // + (a.y b.y, a.y b.x)
    return r;
}
__device__ __forceinline__ v2f cmulk(v2f a, v2f k) { return v2f{a.x * k.x - a.y * k.y, a.x * k.y + a.y * k.x}; }          // by a literal: left to the compiler
__device__ __forceinline__ v2f add_jb(v2f a, v2f b)                            // a + j b = (a.x - b.y, a.y + b.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_jb(v2f a, v2f b)                            // a - j b = (a.x + b.y, a.y - b.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool INV>
__device__ __forceinline__ void dft2(v2f (&x)[2])
{
    const v2f a = x[0] + x[1], b = x[0] - x[1];
    x[0] = a; x[1] = b;
}
template <bool INV>
__device__ __forceinline__ void dft4(v2f (&x)[4])
{
    const v2f s02 = x[0] + x[2], d02 = x[0] - x[2], s13 = x[1] + x[3], d13 = x[1] - x[3];
    x[0] = s02 + s13; x[2] = s02 - s13;
    x[1] = INV ? add_jb(d02, d13) : sub_jb(d02, d13);                          // forward: d02 - j d13
    x[3] = INV ? sub_jb(d02, d13) : add_jb(d02, d13);
}
template <bool INV>
__device__ __forceinline__ void dft5(v2f (&x)[5])
{
    constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749473f, s1 = 0.9510565162951535f, s2 = 0.5877852522924732f;
    const v2f t1 = x[1] + x[4], t2 = x[2] + x[3], t3 = x[1] - x[4], t4 = x[2] - x[3];
    const v2f m1 = x[0] + c1 * t1 + c2 * t2, m2 = x[0] + c2 * t1 + c1 * t2;
    const v2f n1 = s1 * t3 + s2 * t4, n2 = s2 * t3 - s1 * t4;
    x[0] = x[0] + t1 + t2;
    x[1] = INV ? add_jb(m1, n1) : sub_jb(m1, n1);                              // forward: m1 - j n1
    x[4] = INV ? sub_jb(m1, n1) : add_jb(m1, n1);
    x[2] = INV ? add_jb(m2, n2) : sub_jb(m2, n2);
    x[3] = INV ? sub_jb(m2, n2) : add_jb(m2, n2);
}
template <int N, bool INV>
__device__ __forceinline__ void dft_prim(v2f (&x)[N])
{
    static_assert(N == 2 || N == 4 || N == 5, "primitive DFT sizes");
    if constexpr (N == 2) dft2<INV>(x);
    else if constexpr (N == 4) dft4<INV>(x);
    else dft5<INV>(x);
}

// e^{-+ 2 pi i e / N} for the composite sizes: tables of literals, indexed by compile-time constants after unrolling
template <int N, bool INV>
__device__ __forceinline__ v2f tw_const(int e)
{
    static_assert(N == 8 || N == 10 || N == 16 || N == 20, "composite DFT sizes");
    float c, s;
    if constexpr (N == 8) {
        const float C[8] = {1.f, 0.707106781f, 0.f, -0.707106781f, -1.f, -0.707106781f, 0.f, 0.707106781f};
        const float S[8] = {0.f, 0.707106781f, 1.f, 0.707106781f, 0.f, -0.707106781f, -1.f, -0.707106781f};
        c = C[e]; s = S[e];
    } else if constexpr (N == 10) {
        const float C[10] = {1.f, 0.809016994f, 0.309016994f, -0.309016994f, -0.809016994f, -1.f, -0.809016994f, -0.309016994f, 0.309016994f, 0.809016994f};
        const float S[10] = {0.f, 0.587785252f, 0.951056516f, 0.951056516f, 0.587785252f, 0.f, -0.587785252f, -0.951056516f, -0.951056516f, -0.587785252f};
        c = C[e]; s = S[e];
    } else if constexpr (N == 16) {
        const float C[16] = {1.f, 0.923879533f, 0.707106781f, 0.382683432f, 0.f, -0.382683432f, -0.707106781f, -0.923879533f,
                             -1.f, -0.923879533f, -0.707106781f, -0.382683432f, 0.f, 0.382683432f, 0.707106781f, 0.923879533f};
        const float S[16] = {0.f, 0.382683432f, 0.707106781f, 0.923879533f, 1.f, 0.923879533f, 0.707106781f, 0.382683432f,
                             0.f, -0.382683432f, -0.707106781f, -0.923879533f, -1.f, -0.923879533f, -0.707106781f, -0.382683432f};
        c = C[e]; s = S[e];
    } else {
        const float C[20] = {1.f, 0.951056516f, 0.809016994f, 0.587785252f, 0.309016994f, 0.f, -0.309016994f, -0.587785252f, -0.809016994f, -0.951056516f,
                             -1.f, -0.951056516f, -0.809016994f, -0.587785252f, -0.309016994f, 0.f, 0.309016994f, 0.587785252f, 0.809016994f, 0.951056516f};
        const float S[20] = {0.f, 0.309016994f, 0.587785252f, 0.809016994f, 0.951056516f, 1.f, 0.951056516f, 0.809016994f, 0.587785252f, 0.309016994f,
                             0.f, -0.309016994f, -0.587785252f, -0.809016994f, -0.951056516f, -1.f, -0.951056516f, -0.809016994f, -0.587785252f, -0.309016994f};
        c = C[e]; s = S[e];
    }
    return v2f{c, INV ? s : -s};
}

// N = N0 * N1: n = n0 + N0 n1, k = k1 + N1 k0:  N0 DFTs of size N1 over n1, twiddle W_N^{n0 k1}, N1 DFTs of size N0 over n0.  Natural order in and out.
template <int N0, int N1, bool INV>
__device__ __forceinline__ void dft_comp(v2f (&x)[N0 * N1])
{
    v2f s[N1][N0];
#pragma unroll
    for (int n0 = 0; n0 < N0; n0++) {
        v2f t[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; n1++) t[n1] = x[n0 + N0 * n1];
        dft_prim<N1, INV>(t);
#pragma unroll
        for (int k1 = 0; k1 < N1; k1++) s[k1][n0] = (n0 * k1 == 0) ? t[k1] : cmulk(t[k1], tw_const<N0 * N1, INV>(n0 * k1));
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; k1++) {
        dft_prim<N0, INV>(s[k1]);
#pragma unroll
        for (int k0 = 0; k0 < N0; k0++) x[k1 + N1 * k0] = s[k1][k0];
    }
}
template <int N, bool INV>
__device__ __forceinline__ void dft_small(v2f (&x)[N])
{
    static_assert(N == 4 || N == 5 || N == 8 || N == 10 || N == 16 || N == 20, "stripe counts of the fused generator");
    if constexpr (N == 4 || N == 5) dft_prim<N, INV>(x);
    else if constexpr (N == 8) dft_comp<2, 4, INV>(x);
    else if constexpr (N == 10) dft_comp<2, 5, INV>(x);
    else if constexpr (N == 16) dft_comp<4, 4, INV>(x);
    else dft_comp<5, 4, INV>(x);
}

// ---- 1024-point FFT of one wavefront: lane l holds v[a] = x[l + 64 a] on entry and X[l + 64 a] on exit (natural order both ways) ---------------
// index split  n = 64 a + b,  k = p + 16 (u + 16 w):  DFT-16 over a, twiddle W_1024^{b p}, [exchange] DFT-16 over j (b = 4 j + c), twiddle
// W_64^{c u}, [exchange] DFT-4 over c.  xb: this wave's 1088-element LDS slice (rows of 64 padded to 68: both exchanges conflict-free);
// LDS operations of one wave execute in order, so the exchanges need no barrier -- only the compiler must keep them in order.
constexpr int GF_XB = 16 * 68;
#ifndef GF_FFT_WAVES
#define GF_FFT_WAVES 3
#endif
template <bool INV>
__device__ __forceinline__ void fft1024_wave(v2f (&v)[16], const float2 *tw1 /* [16][64]: W_1024^{l p} at [p][l] */, float2 *xb, const float2 *w64, int l)
{
    dft_comp<4, 4, INV>(v);
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const v2f w = lds2(tw1 + p * 64 + l);
        const v2f t = p == 0 ? v[0] : (INV ? cmulc(v[p], w) : cmulv(v[p], w));
        xb[p * 68 + l] = make_float2(t.x, t.y);
    }
    wave_lds_sync();
    const int c = l & 3;
    const float2 *rd = xb + (l >> 2) * 68 + c;
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = lds2(rd + 4 * j);
    dft_comp<4, 4, INV>(v);
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const v2f w = lds2(w64 + ((c * u) & 63));
        const v2f t = u == 0 ? v[0] : (INV ? cmulc(v[u], w) : cmulv(v[u], w));
        xb[l + 64 * u] = make_float2(t.x, t.y);
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float4 lo = *reinterpret_cast<const float4 *>(xb + 4 * (l + 64 * t)), hi = *reinterpret_cast<const float4 *>(xb + 4 * (l + 64 * t) + 2);
        v2f y[4] = {v2f{lo.x, lo.y}, v2f{lo.z, lo.w}, v2f{hi.x, hi.y}, v2f{hi.z, hi.w}};
        dft4<INV>(y);
#pragma unroll
        for (int w = 0; w < 4; w++) v[t + 4 * w] = y[w];
    }
    wave_lds_sync();
}

// ---- tables: W_L^j (forward) and the per-frequency phases of the fibre, laid out [k1][k2] for frequency k1 + N1 k2 ----------------------------
__global__ __launch_bounds__(256) void genf_twiddle_kernel(int L, float2 *__restrict__ T)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    double s, c;
    sincospi(-2.0 * (double)j / (double)L, &s, &c);
    T[j] = make_float2((float)c, (float)s);
}
// (d, ecd): d = e^{j pi f tau_pmd}, ecd = scale e^{j 2 (pi f)^2 tau_cd}  -- the same expressions as gen_disperse_kernel
__global__ __launch_bounds__(256) void genf_phase_kernel(int L, int N1, double fs_over_L, double tau_cd, double tau_pmd, float scale, float4 *__restrict__ H)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int k1 = i >> 10, k2 = i & 1023, f = k1 + N1 * k2;
    const double freq = (double)(f < (L + 1) / 2 ? f : f - L) * fs_over_L;              // np.fft.fftfreq
    const double pf = 3.14159265358979323846 * freq;
    float sc, cc, sd, cd;
    sincosf((float)(2.0 * pf * pf * tau_cd), &sc, &cc);
    sincosf((float)(pf * tau_pmd), &sd, &cd);
    H[i] = make_float4(cd, sd, scale * cc, scale * sc);
}

// ---- pass A ------------------------------------------------------------------------------------------------------------------------------
constexpr int GF_NT = 256, GF_JT = TX_MAXG / 2, GF_SW = 128 + GF_JT + 4;       // taps per phase (max), symbols staged per stripe
constexpr int GF_PARTS = 8;                                                     // partial sums of |sig|^2 per run: [pol][quarter]

template <int N1>
__global__ __launch_bounds__(GF_NT) void genf_tx_kernel(int N_conv, int n_lev, int Lg, int Ls, const float *__restrict__ amp,
                                                        const float *__restrict__ cdf_g, const float2 *__restrict__ g, uint64_t seed,
                                                        uint32_t frame, float2 *__restrict__ sig, int N, int ref_lo, __half *__restrict__ data,
                                                        float *__restrict__ part)
{
    constexpr int Lrow = N1 * 1024;
    __shared__ float2 symL[N1][GF_SW];
    __shared__ float2 gl[2][GF_JT + 1];
    __shared__ float amps[8];
    __shared__ float red[64];
    const int run = blockIdx.z, pol = blockIdx.y, bq = blockIdx.x, tid = threadIdx.x;
    const int JT = (Lg + 1) / 2;                                               // sample s = 2 h + par: sig[s] = sum_j sym[h + par + j] g[Lg-1-par-2j]
    if (tid < n_lev) amps[tid] = amp[tid];
    float cdf7[7];                                                             // uniform (scalar registers): thresholds past n_lev - 1 never fire
#pragma unroll
    for (int i = 0; i < 7; i++) cdf7[i] = i < n_lev - 1 ? cdf_g[(size_t)run * n_lev + i] : 2.0f;
    for (int i = tid; i < 2 * (GF_JT + 1); i += GF_NT) {
        const int par = i / (GF_JT + 1), j = i - par * (GF_JT + 1), k = Lg - 1 - par - 2 * j;
        gl[par][j] = (k >= 0 && j < JT) ? g[k] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int cnt = 128 + JT + 1, PP = (cnt + 1) / 2;                          // symbols a stripe's 256 samples touch (<= GF_SW), in pairs
    __half *dI = data ? data + ((size_t)(run * 2 + pol) * 2 + 0) * N : nullptr, *dQ = dI ? dI + N : nullptr;
    const float rPP = 1.0f / (float)PP;
    for (int q = tid; q < N1 * PP; q += GF_NT) {
        const int n1 = (int)(((float)q + 0.5f) * rPP), pi = q - n1 * PP, n = 512 * n1 + 128 * bq + 2 * pi;   // exact: q < 2^12, PP < 2^8
        const bool last = n1 == N1 - 1 && bq == 3;
        int lv[4];
        {                                                                      // draw_symbol_pair, branch-free
            const Philox4 r = philox4x32_10((uint32_t)n >> 1, run, frame, (uint32_t)(STREAM_SYMBOLS * 2 + pol), (uint32_t)seed, (uint32_t)(seed >> 32));
            const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                lv[c] = 0;
#pragma unroll
                for (int i = 0; i < 7; i++) lv[c] += u[c] >= cdf7[i];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int ne = n + e, m = 2 * pi + e;
            const bool in = ne < N_conv;
            const float aI = amps[lv[2 * e]], aQ = amps[lv[2 * e + 1]];
            symL[n1][m] = in ? make_float2(aI, aQ) : make_float2(0.f, 0.f);
            const int nr = ne - ref_lo;
            if (dI && in && nr >= 0 && nr < N && (m < 128 || last)) {         // each symbol is owned by exactly one (stripe, quarter)
                dI[nr] = __float2half(aI);
                dQ[nr] = __float2half(aQ);
            }
        }
    }
    __syncthreads();
    const int par = tid & 1, hb = (tid >> 1) + par;
    cacc acc[N1];
    {
        const v2f tap = lds2(&gl[par][0]);
#pragma unroll
        for (int n1 = 0; n1 < N1; n1++) cmul(acc[n1], tap, lds2(&symL[n1][hb]));
    }
#pragma unroll 2
    for (int j = 1; j < JT; j++) {
        const v2f tap = lds2(&gl[par][j]);
#pragma unroll
        for (int n1 = 0; n1 < N1; n1++) cmac(acc[n1], tap, lds2(&symL[n1][hb + j]));
    }
    v2f x[N1];
    float pw = 0.f;
    const int n2 = 256 * bq + tid;
#pragma unroll
    for (int n1 = 0; n1 < N1; n1++) {
        const v2f v = cfin2(acc[n1]);
        x[n1] = 1024 * n1 + n2 < Ls ? v : v2f{0.f, 0.f};                       // zero padding of the row (linear, not circular, filtering)
        pw += x[n1].x * x[n1].x + x[n1].y * x[n1].y;
    }
    dft_small<N1, false>(x);
    float2 *o = sig + (size_t)(run * 2 + pol) * Lrow + n2;
#pragma unroll
    for (int k1 = 0; k1 < N1; k1++) o[k1 * 1024] = make_float2(x[k1].x, x[k1].y);
    if (part) {                                                                // fixed-order block sum: bitwise reproducible
        block_reduce3<GF_NT>(pw, 0.f, 0.f, red);
        if (tid == 0) part[(size_t)(run * 2 + pol) * 4 + bq] = red[0];
    }
}

// ---- pass B ------------------------------------------------------------------------------------------------------------------------------
template <int N1>
__global__ __launch_bounds__(256, GF_FFT_WAVES) void genf_fft_kernel(int R, int rpw, const float2 *__restrict__ T, const float4 *__restrict__ H, float2 E00,
                                                       float2 E01, float2 E11, const float *__restrict__ theta, float2 *__restrict__ sig)
{
    constexpr int Lrow = N1 * 1024;
    __shared__ float2 xbs[4][GF_XB];
    __shared__ float2 w64[64];
    __shared__ float2 tw1[16 * 64];                                            // W_1024^{l p} at [p][l]: lane-contiguous reads
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, k1 = blockIdx.y;
    if (tid < 64) w64[tid] = T[N1 * 16 * tid];                                 // W_64^t
    __shared__ float2 twio[16 * 64];                                           // W_L^{k1 (l + 64 r)} at [r][l]: the seam between the N1- and the 1024-point stage
    for (int i = tid; i < 16 * 64; i += 256) {
        tw1[i] = T[N1 * (((i & 63) * (i >> 6)) & 1023)];
        twio[i] = T[k1 * ((i & 63) + 64 * (i >> 6))];                          // k1 * 1023 < L
    }
    __syncthreads();
    float2 *xb = xbs[wv];
    const float4 *Hr = H + k1 * 1024 + l;
    const int r0 = (blockIdx.x * 4 + wv) * rpw, r1 = min(R, r0 + rpw);
    for (int run = r0; run < r1; run++) {
        float st, ct;
        sincosf(theta[run], &st, &ct);
        const v2f c00 = {ct * E00.x, ct * E00.y}, c01 = {st * E01.x, st * E01.y}, c11 = {ct * E11.x, ct * E11.y};
        float2 *p0 = sig + (size_t)(run * 2) * Lrow + k1 * 1024 + l, *p1 = p0 + Lrow;
        v2f a[16], b[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float2 u = p0[64 * r], w = p1[64 * r];
            a[r] = v2f{u.x, u.y};
            b[r] = v2f{w.x, w.y};
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const v2f tw = lds2(twio + 64 * r + l);
            a[r] = cmulv(a[r], tw);
            b[r] = cmulv(b[r], tw);
        }
        fft1024_wave<false>(a, tw1, xb, w64, l);
        fft1024_wave<false>(b, tw1, xb, w64, l);
        wave_lds_sync();                                                       // (compiler fence: the table loads below stay below the transforms)
#pragma unroll
        for (int r4 = 0; r4 < 16; r4 += 4) {                                   // frequency k1 + N1 (l + 64 r), four at a time
            float4 h[4];
#pragma unroll
            for (int i = 0; i < 4; i++) h[i] = Hr[64 * (r4 + i)];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r4 + i;
                const v2f d = {h[i].x, h[i].y}, ecd = {h[i].z, h[i].w};
                const v2f u = ct * a[r] + st * b[r], w = ct * b[r] - st * a[r];
                const v2f p = cmulv(u, d), q = cmulc(w, d);
                a[r] = cmulv(cmulv(c00, p) - cmulv(c01, q), ecd);
                b[r] = cmulv(cmulv(c01, p) + cmulv(c11, q), ecd);
            }
            wave_lds_sync();
        }
        fft1024_wave<true>(a, tw1, xb, w64, l);
        fft1024_wave<true>(b, tw1, xb, w64, l);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const v2f tw = lds2(twio + 64 * r + l);
            const v2f ao = cmulc(a[r], tw), bo = cmulc(b[r], tw);
            p0[64 * r] = make_float2(ao.x, ao.y);
            p1[64 * r] = make_float2(bo.x, bo.y);
        }
    }
}

// ---- pass C ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void genf_noise(float sigma, uint32_t w0, uint32_t w1, float &nI, float &nQ)
{
    float sn, cs;
    const float rad = sigma * sqrtf(-2.0f * __logf(u01(w0)));
    __sincosf(6.283185307179586f * u01(w1), &sn, &cs);
    nI = rad * cs;
    nQ = rad * sn;
}

template <int N1>
__global__ __launch_bounds__(GF_NT) void genf_finish_kernel(int Lout, const float *__restrict__ sigma_src, int sigma_stride, uint64_t seed, uint32_t frame,
                                                            const float2 *__restrict__ sig, float *__restrict__ rx, float *__restrict__ sigma_out)
{
    constexpr int Lrow = N1 * 1024;
    const int run = blockIdx.z, pol = blockIdx.y, tid = threadIdx.x, m1 = blockIdx.x * GF_NT + tid, par = tid & 1;
    const float sigma = sigma_src[(size_t)run * sigma_stride];
    if (sigma_out && pol == 0 && blockIdx.x == 0 && tid == 0) sigma_out[run] = sigma;
    const float2 *s = sig + (size_t)(run * 2 + pol) * Lrow + m1;
    v2f x[N1];
#pragma unroll
    for (int k1 = 0; k1 < N1; k1++) {
        const float2 t = s[k1 * 1024];
        x[k1] = v2f{t.x, t.y};
    }
    dft_small<N1, true>(x);                                                    // x[m2] = sample m1 + 1024 m2
    float *rI = rx + ((size_t)(run * 2 + pol) * 2 + 0) * Lout, *rQ = rI + Lout;
    // noise word j = sample >> 1 serves samples 2j (words x, y) and 2j + 1 (z, w): the lanes of a pair (m1 even, m1 odd) share it.  For the
    // stripes (2t, 2t + 1) the even lane draws word j(2t), the odd lane j(2t + 1), and they swap the halves the other one needs.
    const uint32_t hj = (uint32_t)(m1 >> 1), k0 = (uint32_t)seed, k1_ = (uint32_t)(seed >> 32), strm = (uint32_t)(STREAM_NOISE * 2 + pol);
#pragma unroll
    for (int t = 0; t < N1 / 2; t++) {
        const Philox4 r = philox4x32_10(hj + 512u * (uint32_t)(2 * t + par), run, frame, strm, k0, k1_);
        const uint32_t own0 = par ? r.z : r.x, own1 = par ? r.w : r.y, snd0 = par ? r.x : r.z, snd1 = par ? r.y : r.w;
        const uint32_t rcv0 = (uint32_t)__shfl_xor((int)snd0, 1), rcv1 = (uint32_t)__shfl_xor((int)snd1, 1);
        const uint32_t a0 = par ? rcv0 : own0, a1 = par ? rcv1 : own1, b0 = par ? own0 : rcv0, b1 = par ? own1 : rcv1;   // stripes 2t, 2t + 1
        float nI, nQ;
        const int sa = m1 + 1024 * (2 * t), sb = sa + 1024;
        genf_noise(sigma, a0, a1, nI, nQ);
        if (sa < Lout) { rI[sa] = x[2 * t].x + nI; rQ[sa] = x[2 * t].y + nQ; }
        genf_noise(sigma, b0, b1, nI, nQ);
        if (sb < Lout) { rI[sb] = x[2 * t + 1].x + nI; rQ[sb] = x[2 * t + 1].y + nQ; }
    }
    if constexpr (N1 & 1) {
        const Philox4 r = philox4x32_10(hj + 512u * (uint32_t)(N1 - 1), run, frame, strm, k0, k1_);
        float nI, nQ;
        const int sa = m1 + 1024 * (N1 - 1);
        genf_noise(sigma, par ? r.z : r.x, par ? r.w : r.y, nI, nQ);
        if (sa < Lout) { rI[sa] = x[N1 - 1].x + nI; rQ[sa] = x[N1 - 1].y + nQ; }
    }
}

}  // namespace vaeq
