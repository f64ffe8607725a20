// Feasibility probe (round 5): fp32-accurate contractions on the bf16 matrix cores by operand splitting (a = a1 + a2 + a3, three bf16 terms = 24 significand bits;
// products of bf16 terms are exact in fp32; 6 of the 9 cross terms carry everything down to 2^-24 relative).  Question: what does the SAME contraction depth cost
// as v_mfma_f32_16x16x4_f32 (24 instructions = K 96) vs v_mfma_f32_16x16x32_bf16 (3 K-steps x 6 or x 9 products), with NV fp32 VALU operations beside it
// (on the fp32 MFMA they do not overlap: profiles/r1_ubench_mfma_valu.txt)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bf16x3.hip -o /tmp/ubench_bf16x3 && /tmp/ubench_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>      // MODE 0: 24 fp32 MFMAs; 6 / 9: 3 K-steps x MODE bf16 MFMAs
__global__ void k(float* out, int iters, float seed) {
    f32x4 acc[2] = {{seed, seed, seed, seed}, {seed, seed, seed, seed}};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(seed + i); pb[i] = (__bf16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    constexpr int NM = MODE == 0 ? 24 : 3 * MODE;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (MODE == 0) acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 1], 0, 0, 0);
            else acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[m & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < (NV * (m + 1)) / NM - (NV * m) / NM; ++j) { const int q = (m * 3 + j) & 7; v[q] = fmaf(v[q], 1.0001f, 0.5f); }
        }
    }
    float s = acc[0][0] + acc[1][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NV> double run(int wps, float* d) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / (iters * wps);       // shader cycles (at 2.4 GHz) per K = 96 block per wave slot
}
template <int NV> void row(float* d) {
    for (int wps : {1, 2, 4})
        printf("NV=%3d waves/SIMD=%d : fp32 MFMA x24 %6.0f cyc | bf16 x6 (18 MFMAs) %6.0f | bf16 x9 (27 MFMAs) %6.0f\n", NV, wps, run<0, NV>(wps, d), run<6, NV>(wps, d), run<9, NV>(wps, d));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
    row<0>(d); row<24>(d); row<60>(d); row<96>(d); row<144>(d);
    return 0;
}
