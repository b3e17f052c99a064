// vaeq_validate.h -- the part of the AWGN validation passes that is common to the VAE-LE and the VAE-NN equalizer
// (func_VAELE_MQAM_shaping.py / func_VAENN_MQAM.py have identical find_shift and SER_q): shift search on E_q[x_I] of the first
// 1000 symbols, then the 4-rotation SER on nibble-packed hard decisions held in LDS.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "vaeq_common.h"

namespace vaeq {

constexpr int VAL_NE = 1000, VAL_MAXSHIFT = 32;

// decs[N]: I decision in the low, Q in the high nibble; E[NE]: E_q[x_I]; data: TX reference [2][N] fp16 of this run.
// find_shift (:188-204 of either file): corr[i] = <tx[:NE], roll(E, i - half)>, roll(E, s)[n] = E[(n - s) mod NE].
// SER_q (:97-123) on q[:, 11+sh : -11] vs data[:, 11 : -11-sh], minimum over the four quadrant rotations.
template <int NT, int NLEV>
__device__ __forceinline__ void validate_tail(int N, int n_shift, const unsigned char *decs, const float *E, int NE, const __half *data,
                                              float *red, float (*corr)[VAL_MAXSHIFT], int *sh_s, float *ser_out, int *shift_out)
{
    const int tid = threadIdx.x;
    const __half *tI = data, *tQ = tI + N;
    const int half_ = n_shift / 2;
    for (int i = 0; i < n_shift; i++) {
        float cI = 0.f, cQ = 0.f;
        for (int n = tid; n < NE; n += NT) {
            int m = n - (i - half_);
            m = m < 0 ? m + NE : (m >= NE ? m - NE : m);
            const float e = E[m];
            cI = fmaf(__half2float(tI[n]), e, cI);
            cQ = fmaf(__half2float(tQ[n]), e, cQ);
        }
        __syncthreads();
        block_reduce3<NT>(cI, cQ, 0.f, red);
        if (tid == 0) { corr[0][i] = fabsf(red[0]); corr[1][i] = fabsf(red[1]); }
    }
    if (tid == 0) {
        int aI = 0, aQ = 0;
        for (int i = 1; i < n_shift; i++) {
            if (corr[0][i] > corr[0][aI]) aI = i;
            if (corr[1][i] > corr[1][aQ]) aQ = i;
        }
        int sh = half_ - aI;
        if (!(corr[0][aI] >= (float)(0.02 * (double)N)) && corr[1][aQ] >= corr[0][aI]) sh = half_ - aQ;
        *sh_s = sh;
        if (shift_out) *shift_out = sh;
    }
    __syncthreads();
    const int sh = *sh_s, len = N - 22 - sh, K = NLEV - 1;
    const float scale = 0.5f * (float)K;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int j = tid; j < len; j += NT) {
        const int dd = decs[11 + sh + j], dI = dd & 15, dQ = dd >> 4;
        const int aI = (int)rintf(__fadd_rn(__fmul_rn(scale, __half2float(tI[11 + j])), scale));
        const int aQ = (int)rintf(__fadd_rn(__fmul_rn(scale, __half2float(tQ[11 + j])), scale));
        c0 += (aI != dI) | (aQ != dQ);
        c1 += (aI != K - dI) | (aQ != K - dQ);
        c2 += (aI != K - dQ) | (aQ != dI);
        c3 += (aI != dQ) | (aQ != K - dI);
    }
    __syncthreads();
    block_reduce3<NT>(c0, c1, c2, red);
    const float r0 = red[0], r1 = red[1], r2 = red[2];
    __syncthreads();
    block_reduce3<NT>(c3, 0.f, 0.f, red);
    if (tid == 0) *ser_out = fminf(fminf(r0, r1), fminf(r2, red[0])) / (float)len;
}

}  // namespace vaeq
