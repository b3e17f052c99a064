// Micro-benchmark for the "one symbol per lane, three waves per SIMD" question (VERDICT r2 item 4 iii): the DP kernel's convolution-shaped phases
// (62 % of a step) as they would run in either mapping, isolated: same LDS layout (8-byte reads, conflict-free strides, broadcast tap reads), same
// two-deep software pipeline, same packed-FMA chains.
//   pair   : a lane owns a symbol PAIR  -> per group of 4 taps 6 sample reads + 8 tap reads feed 32 v_pk_fma_f32 (0.44 reads per FMA); 254 VGPRs in
//            the real kernel => 2 waves per SIMD (8 wavefronts per CU)
//   single : a lane owns ONE symbol     -> per group of 4 taps 4 sample reads + 8 tap reads feed 16 v_pk_fma_f32 (0.75 reads per FMA); the halved per-lane
//            state is what would let 3 waves per SIMD fit (12 wavefronts per CU)
// Both process the same number of symbol-taps per "unit"; the figure of merit is symbol-taps per ns per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tapshape tapshape.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v2f lds_cv2f;
__device__ __forceinline__ v2f lds2(const float2 *p) { return *(lds_cv2f *)p; }
struct cacc { v2f a, b; };
__device__ __forceinline__ void cmac(cacc &c, v2f t, v2f v) { c.a += t.x * v; c.b += t.y * v; }

// MODE 0 = pair, 1 = single.  LDS: x[4 phases][72] float2 (stride 72: 16 (mod 64) dwords apart), taps[32] float4
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters)
{
    extern __shared__ float4 sm4[];
    float2 *xs = reinterpret_cast<float2 *>(sm4);
    float2 *tp = xs + 4 * 72;
    const int l = threadIdx.x;
    for (int i = l; i < 4 * 72 + 64; i += 64) xs[i] = make_float2(0.001f * i, 1.0f - 0.002f * i);
    __syncthreads();
    constexpr int NA = MODE == 0 ? 8 : 4;                 // complex accumulators: [sym][o][even/odd tap]
    cacc acc[NA];
    for (int i = 0; i < NA; i++) acc[i] = cacc{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    const float2 *xl = xs + (l & 31);                     // (offsets stay inside the 72-slot rows for g < 6)
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
        for (int g = 0; g < 6; g++) {                     // 6 groups of 4 taps = the M = 25 FIR's tap loop
            const float2 *xg = xl + g;
            v2f t[8];
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = lds2(tp + 8 * (g & 3) + i);      // (re, im) of o = 0 / o = 1 for 4 taps: broadcast reads
            if (MODE == 0) {
                const v2f x0 = lds2(xg), x1 = lds2(xg + 72), x2 = lds2(xg + 144), x3 = lds2(xg + 216), x4 = lds2(xg + 1), x5 = lds2(xg + 73);
                cmac(acc[0], t[0], x0); cmac(acc[1], t[1], x0); cmac(acc[2], t[0], x2); cmac(acc[3], t[1], x2);
                cmac(acc[4], t[2], x1); cmac(acc[5], t[3], x1); cmac(acc[6], t[2], x3); cmac(acc[7], t[3], x3);
                cmac(acc[0], t[4], x2); cmac(acc[1], t[5], x2); cmac(acc[2], t[4], x4); cmac(acc[3], t[5], x4);
                cmac(acc[4], t[6], x3); cmac(acc[5], t[7], x3); cmac(acc[6], t[6], x5); cmac(acc[7], t[7], x5);
            } else {
                const v2f x0 = lds2(xg), x1 = lds2(xg + 72), x2 = lds2(xg + 144), x3 = lds2(xg + 216);
                cmac(acc[0], t[0], x0); cmac(acc[1], t[1], x0);
                cmac(acc[2], t[2], x1); cmac(acc[3], t[3], x1);
                cmac(acc[0], t[4], x2); cmac(acc[1], t[5], x2);
                cmac(acc[2], t[6], x3); cmac(acc[3], t[7], x3);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < NA; i++) s += acc[i].a.x + acc[i].a.y + acc[i].b.x + acc[i].b.y;
    out[blockIdx.x * 64 + l] = s;
}

template <int MODE>
static void run(const char *name, int waves_per_simd, float *out)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int per_cu = 4 * waves_per_simd, blocks = p.multiProcessorCount * per_cu * 4;
    const size_t lds = (size_t)(160 * 1024 / per_cu) - 512;            // LDS sized so that exactly per_cu workgroups of one wave are resident per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double symtaps = (double)blocks * 64 * (MODE == 0 ? 2 : 1) * 24.0 * iters;     // symbols x taps (each = 2 complex MACs: o = 0, 1)
    printf("%-7s %d waves/SIMD: %8.3f ms  %8.1f symbol-taps per ns per chip  (%.2f per ns per CU)\n", name, waves_per_simd, ms, symtaps / (ms * 1e6),
           symtaps / (ms * 1e6) / p.multiProcessorCount);
}

int main()
{
    float *out;
    hipMalloc(&out, 64 * 4 * 256 * 64 * 4);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("pair", 2, out);
        run<1>("single", 3, out);
        run<1>("single", 2, out);
        run<0>("pair", 3, out);
    }
    return 0;
}
