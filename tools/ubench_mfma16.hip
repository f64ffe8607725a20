// Issue cost of the 16-bit MFMA shapes on gfx950 (round 6): is a K = 16 step (v_mfma_f32_16x16x16_f16) half the price of a K = 32 step
// (v_mfma_f32_16x16x32_f16)?  One wave per SIMD on every CU, N independent accumulator chains, cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
template <int K32>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[4]; for (int c = 0; c < 4; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    half8 a8, b8; half4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 0.001f + e); b8[e] = (_Float16)(0.5f + e); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (K32) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[m & 3], 0, 0, 0);
            else acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[m & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* d; unsigned long long* c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 4000;
    for (int which = 0; which < 2; ++which) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (which) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, c, iters); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, c, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f s_memtime ticks per MFMA (one wave per SIMD, 4 chains)\n", which ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x16_f16", (double)h / (iters * 16.0));
    }
    return 0;
}
