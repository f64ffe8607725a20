// Term splits of fp32 operands for the bf16 / fp16 matrix cores (round 5; range rule round 6), shared by fused_mlp_x3.h, fused_attn.h, fused_rowgemm.h, fused_deembed.h.
//   NT = 3: a = a1 + a2 + a3 EXACTLY, a_i bf16 (8-bit significands, round to nearest at each step, exact fp32 residuals); six cross products carry a . b to 2^-24.
//           bf16 has fp32's exponent range: no scaling, no range condition.
//   NT = 2: a ~ a1 + a2, a1 = fp16(a), a2 = fp16(a - a1) (11-bit significands: |a - a1 - a2| <= 2^-22 |a|, typically 2^-24); THREE cross products a1b1 + a1b2 + a2b1 on
//           v_mfma_f32_16x16x32_f16 - half the matrix instructions of NT = 3.  Not exact: the truncation measures 9.5e-8 of the mean result magnitude (numpy, exact
//           accumulation) next to 1.6e-7 ... 4.5e-7 of fp32 accumulation rounding for K = 48 ... 1536; on the device every TransformerLayer stays at 0.6-0.9x the reference's
//           own float32 error against float64 (tests/test_gpu_parity.py test_layer_accuracy_against_fp64).
// RANGE RULE of NT = 2 (fp16 has a 5-bit exponent: overflow at 65520, low terms subnormal below 2^-3).  EVERY operand is multiplied by a power of two before it is split
// and the accumulator is multiplied back by the inverse - powers of two commute with fp32 rounding, so the scaling itself changes no bit:
//   * weights: per matrix, max |w| 2^k in [2^13, 2^14) (x2_scale of absmax_bits_kernel, when the image is packed);
//   * LayerNorm outputs (fc1, Q / K / V, PatchMerge / PatchSplit inputs): per layer, from the bound |xn_c| <= max |gamma| sqrt(C) + max |beta| that the normalisation
//     guarantees for ANY input (sum z^2 = C), brought into [2^13, 2^14) (ln_act_scale, when the image is packed);
//   * GELU outputs (fc2 input): per layer, from |gelu(h)| <= |h| <= max_row ||w1_row||_2 sqrt(C) (max |gamma| + max |beta|) + max |b1| (Cauchy-Schwarz), likewise;
//   * de-embedding input (raw decoder tokens, no norm in front: scale.py:73-81): per workgroup tile, from the tile's own max |x| (fused_deembed.h).
// With these no finite input can overflow a high term, and the low term of every value within 2^-10 of its bound is a normal fp16 number.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_bf16.h"

namespace escx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// one 16 x 16 x 32 step on fragments held as 16 raw bytes per lane
template <int NT>
__device__ __forceinline__ f32x4 mma_x(bf16x8 w, bf16x8 x, f32x4 c) {
    if constexpr (NT == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, w), __builtin_bit_cast(half8, x), c, 0, 0, 0);
}

// eight fp32 values (times an exact power-of-two scale for NT = 2) -> NT term fragments
template <int NT>
__device__ __forceinline__ void split_terms(const float (&v)[8], bf16x8 (&out)[NT], float scale = 1.0f) {
#pragma clang fp contract(off)
    if constexpr (NT == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 a1 = (__bf16)v[e];
            const float r1 = v[e] - (float)a1;
            const __bf16 a2 = (__bf16)r1;
            const float r2 = r1 - (float)a2;
            out[0][e] = a1; out[1][e] = a2; out[2][e] = (__bf16)r2;
        }
    } else {
        half8 h0, h1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = v[e] * scale;
            const _Float16 a1 = (_Float16)a;
            h0[e] = a1; h1[e] = (_Float16)(a - (float)a1);
        }
        out[0] = __builtin_bit_cast(bf16x8, h0); out[1] = __builtin_bit_cast(bf16x8, h1);
    }
}

// power of two that brings a bound B into [2^13, 2^14) (1 for 0 / inf / NaN): the activation scales of the range rule
__device__ __forceinline__ float x2_scale(unsigned absmax_bits);
__device__ __forceinline__ float act_pow2_scale(float bound) { return x2_scale(__float_as_uint(fabsf(bound))); }
// bound on a LayerNorm output over n channels (C real ones): max |gamma| sqrt(C) + max |beta|; *gb = max |gamma| + max |beta| for the fc1 bound.  One thread, n <= 768.
__device__ inline float ln_out_bound(const float* gamma, const float* beta, int n, int C, float* gb = nullptr) {
    float mg = 0.f, mb = 0.f;
    for (int i = 0; i < n; ++i) { mg = fmaxf(mg, fabsf(gamma[i])); mb = fmaxf(mb, fabsf(beta[i])); }
    if (gb) *gb = mg + mb;
    return mg * sqrtf((float)C) + mb;
}

// power of two that brings max |w| into [2^13, 2^14) (1 for an all-zero matrix): bits of max |w| -> scale
__device__ __forceinline__ float x2_scale(unsigned absmax_bits) {
    const int ex = (int)((absmax_bits >> 23) & 0xff);       // biased exponent of max |w|
    if (ex == 0 || ex == 0xff) return 1.0f;
    const int e = 127 + 13 - (ex - 127);                    // biased exponent of the scale; clamped so that scale AND 1 / scale stay normal fp32 numbers
    return __uint_as_float((unsigned)(e < 24 ? 24 : (e > 230 ? 230 : e)) << 23);
}

// bits of max |w[i]| -> *out (atomicMax: non-negative floats order like their bit patterns; the maximum does not depend on the order of the updates)
static __global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(w[i])));
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// the same over the tiles of a fragment stream whose kind (tile index mod tpg) is selected by `mask` (attention streams: Q / K / V tiles, V tiles, projection tiles)
static __global__ __launch_bounds__(256) void absmax_tiles_bits_kernel(const float* __restrict__ w, long long n, int tile_floats, int tpg, unsigned mask, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        if ((mask >> (int)((i / tile_floats) % tpg)) & 1) m = max(m, __float_as_uint(fabsf(w[i])));
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// bits of max_r sum_c w[r][c]^2 -> *out (one wave per row; the fc1 bound of the range rule)
static __global__ __launch_bounds__(256) void rownorm2_max_bits_kernel(const float* __restrict__ w, int rows, int cols, int ld, unsigned* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float s = 0.f;
    if (r < rows) for (int c = lane; c < cols; c += 64) { const float v = w[(size_t)r * ld + c]; s += v * v; }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && r < rows) atomicMax(out, __float_as_uint(s));
}

// the cross terms (weight term i, activation term j), smallest first
#define ESCX_X2_TERMS(M) M(0, 1) M(1, 0) M(0, 0)

}  // namespace escx
