// Micro-benchmark: issue cost of the 32 x 32 -> 64-bit multiplies Philox4x32 is made of (v_mad_u64_u32 vs v_mul_lo_u32 + v_mul_hi_u32) and of one
// whole Philox4x32-10 call, per wavefront, at 1 / 2 / 4 / 8 waves per SIMD.     build: hipcc --offload-arch=gfx950 -O3 -o intmul intmul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, long long *cyc)
{
    uint32_t a[8], m = 0xD2511F53u + threadIdx.x;
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    uint64_t w[8];
    for (int i = 0; i < 8; i++) w[i] = a[i];
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)a[i];
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {                                  // 16 independent v_mad_u64_u32
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(m), "v"(a[i]) : "vcc");
        }
        if (MODE == 1) {                                  // 16 v_mul_lo_u32
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
        if (MODE == 2) {                                  // 16 v_mul_hi_u32
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
        if (MODE == 3) {                                  // 16 v_fma_f32 (reference: full rate)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        }
        if (MODE == 4) {                                  // one Philox4x32-10 call (dependent chain), the compiler's code
            uint32_t c0 = a[0], c1 = a[1], c2 = a[2], c3 = a[3], k0 = a[4], k1 = a[5];
            for (int r = 0; r < 10; r++) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
                const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
                c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
            }
            a[0] = c0; a[1] = c1; a[2] = c2; a[3] = c3;
        }
    }
    long long t1 = clock64();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32) + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int per_iter)
{
    uint32_t *out; long long *cyc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 2048 * 8);
    for (int wps : {1, 2, 4, 8}) {                        // waves per SIMD: 256 CUs x 4 SIMDs; block = 4 waves = one per SIMD of a CU
        const int blocks = 256 * wps, iters = 2000;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ghz = 2.4, simd_cycles = ms * 1e-3 * ghz * 1e9;
        printf("%-34s %d waves/SIMD: %7.2f SIMD cycles per wave-instruction-group (%d per iteration) -> %6.2f cycles each per wave (at %.1f GHz)\n", name, wps,
               simd_cycles / iters / wps, per_iter, simd_cycles / iters / wps / per_iter, ghz);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<3>("v_fma_f32 x16", 16);
    run<0>("v_mad_u64_u32 x16", 16);
    run<1>("v_mul_lo_u32 x16", 16);
    run<2>("v_mul_hi_u32 x16", 16);
    run<4>("Philox4x32-10 call", 1);
    return 0;
}
