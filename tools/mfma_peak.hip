// Pure-MFMA ceiling: how many TFLOP/s does a loop of independent f32 MFMAs reach with 1..3 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void peak_kernel(int iters, float* out) {
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    if (KIND == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int bpc = 1; bpc <= 4; ++bpc) {       // blocks of 4 waves per CU = waves per SIMD
            const int grid = 256 * bpc;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) peak_kernel<0><<<grid, 256>>>(iters, out); else peak_kernel<1><<<grid, 256>>>(iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // flops: 32x32x2: 4096 per MFMA, 64 per iter per wave; 16x16x4: 2048 per MFMA, 128 per iter
            const double flop = (double)grid * 4 * iters * (kind == 0 ? 64.0 * 4096 : 128.0 * 2048);
            printf("%s waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "32x32x2" : "16x16x4", bpc, ms, flop / ms / 1e9);
        }
    return 0;
}
