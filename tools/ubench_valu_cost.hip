#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// KIND: 0 fma, 1 pk_fma (float2), 2 exp2, 3 max, 4 pk_add, 5 cndmask(select), 6 permlane32_swap, 7 mov_dpp, 8 salu add, 9 rcp
template <int KIND, int NV>
__global__ void k(float* out, int iters, float seed) {
    f32x4 acc[2] = {{seed, seed, seed, seed}, {seed, seed, seed, seed}};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8]; f32x2 p[8]; int sacc = iters;
    for (int i = 0; i < 8; ++i) { v[i] = seed + i; p[i] = {seed + i, seed - i}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < (NV * (m + 1)) / 24 - (NV * m) / 24; ++j) {
                const int q = (m * 3 + j) & 7;
                if (KIND == 0) v[q] = fmaf(v[q], 1.0001f, 0.5f);
                if (KIND == 1) p[q] = p[q] * f32x2{1.0001f, 1.0002f} + f32x2{0.5f, 0.25f};
                if (KIND == 2) v[q] = __builtin_amdgcn_exp2f(v[q]);
                if (KIND == 3) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[q]) : "v"(v[q]), "v"(v[(q + 1) & 7]));
                if (KIND == 4) p[q] = p[q] + p[(q + 1) & 7];
                if (KIND == 5) v[q] = (v[(q + 3) & 7] > 0.5f) ? v[q] : v[(q + 1) & 7];
                if (KIND == 6) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[q]), __float_as_uint(v[(q + 1) & 7]), false, false); v[q] = __uint_as_float(r[0]); v[(q + 1) & 7] = __uint_as_float(r[1]); }
                if (KIND == 7) v[q] = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v[q]), 0x128, 0xf, 0xf, true));
                if (KIND == 8) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
                if (KIND == 9) v[q] = __builtin_amdgcn_rcpf(v[q]);
            }
        }
    }
    float s = acc[0][0] + acc[1][3] + (float)sacc;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NV> double run(int wps, float* d) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / (iters * wps);       // cycles per 24-MFMA block per wave slot (at 2.4 GHz)
}
template <int KIND> void kind(const char* name, float* d) {
    const double base = run<KIND, 0>(2, d), c48 = run<KIND, 48>(2, d), c96 = run<KIND, 96>(2, d);
    printf("%-16s base %6.0f cyc/24 MFMA; +48 ops: %6.0f (%.2f cyc/op); +96 ops: %6.0f (%.2f cyc/op)\n", name, base, c48, (c48 - base) / 48, c96, (c96 - base) / 96);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
    kind<0>("v_fma_f32", d); kind<1>("v_pk_fma_f32", d); kind<2>("v_exp_f32", d); kind<3>("v_max_f32", d); kind<4>("v_pk_add_f32", d);
    kind<5>("cmp+cndmask", d); kind<6>("permlane32_swap", d); kind<7>("mov_dpp", d); kind<8>("s_add_i32", d); kind<9>("v_rcp_f32", d);
    return 0;
}
