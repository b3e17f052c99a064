// Micro-benchmark: do f32 MFMAs co-issue with v_pk_fma_f32 on gfx950?  (decides whether moving the DP kernel's FMAs to the matrix pipe can pay)
// build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc)
{
    v2f a[8], x = {1.0f + threadIdx.x, 2.0f}, t = {0.5f, 0.25f};
    v4f d[8];
    v16f e[2];
    for (int i = 0; i < 8; i++) { a[i] = v2f{0.f, (float)i}; d[i] = v4f{0, 0, 0, (float)i}; }
    for (int i = 0; i < 2; i++) for (int j = 0; j < 16; j++) e[i][j] = (float)j;
    float ma = threadIdx.x * 0.01f, mb = 1.0f - threadIdx.x * 0.001f;
    v2f a2[8];
    for (int i = 0; i < 8; i++) a2[i] = v2f{1.f, (float)i};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 7) {                                  // 16 pk_fma on 16 independent accumulators
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t), "v"(x));
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(t), "v"(x));
        }
        if (MODE == 8) {                                  // 16 pk_fma on 4 accumulators
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t), "v"(x));
        }
        if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6) {       // 16 pk_fma (8 accumulators x 2)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t), "v"(x));
        }
        if (MODE == 1 || MODE == 2) {                    // 8 mfma 4x4x1 (same MAC count as 16 pk_fma)
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, mb, d[i], 0, 0, 0);
        }
        if (MODE == 3 || MODE == 4) {                    // 2 mfma 16x16x4 (2 x 1024 MACs = same as 16 pk_fma x 128 MACs)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                v4f c = {e[i][0], e[i][1], e[i][2], e[i][3]};
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, c, 0, 0, 0);
                e[i][0] = c[0]; e[i][1] = c[1]; e[i][2] = c[2]; e[i][3] = c[3];
            }
        }
        if (MODE == 5 || MODE == 6) {                    // 2 mfma 16x16x1 4B (4 blocks; 2 x 1024 MACs)
#pragma unroll
            for (int i = 0; i < 2; i++) e[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(ma, mb, e[i], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + d[i][0] + d[i][3] + a2[i].x + a2[i].y;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 16; j++) s += e[i][j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd)
{
    float *out; long long *cyc, h;
    int blocks = 256 * 4 * waves_per_simd, iters = 20000;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, iters, cyc); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s waves/SIMD %d: %8.3f ms  %7.1f ns/iter  (clock64 ticks/iter %.1f)\n", name, waves_per_simd, ms, ms * 1e6 / iters, (double)h / iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 2; w++) {
        run<0>("16 pk_fma (8 chains)", w);
        run<7>("16 pk_fma (16 chains)", w);
        run<8>("16 pk_fma (4 chains)", w);
        run<1>("8 mfma_4x4x1", w);
        run<2>("16 pk_fma + 8 mfma_4x4x1", w);
        run<3>("2 mfma_16x16x4", w);
        run<4>("16 pk_fma + 2 mfma_16x16x4", w);
        run<5>("2 mfma_16x16x1(4B)", w);
        run<6>("16 pk_fma + 2 mfma_16x16x1(4B)", w);
    }
    return 0;
}
