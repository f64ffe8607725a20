#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// NV VALU fmas per 24 MFMAs; CH independent MFMA chains; waves per block = blockDim/64
template <int NV, int CH>
__global__ void k(float* out, int iters, float seed) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = {seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % CH], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < (NV * (m + 1)) / 24 - (NV * m) / 24; ++j) { const int q = (m * 3 + j) & 7; v[q] = fmaf(v[q], 1.0001f, 0.5f); }
        }
    }
    float s = 0; for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int CH> void run(int waves_per_simd, float* d) {
    const int iters = 2000;
    const int threads = 256;            // 4 waves per block, 1 per SIMD
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, CH>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, CH>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 24 * 2048.0;
    printf("NV=%3d CH=%d waves/SIMD=%d : %7.1f TFLOP/s (%.3f ms)\n", NV, CH, waves_per_simd, flops / ms / 1e9, ms);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
    for (int w : {1, 2, 4}) {
        run<0, 2>(w, d); run<0, 4>(w, d); run<24, 2>(w, d); run<60, 2>(w, d); run<60, 4>(w, d); run<96, 2>(w, d); run<144, 2>(w, d);
    }
    return 0;
}
