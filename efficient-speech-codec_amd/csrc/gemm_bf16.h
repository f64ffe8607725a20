// bf16-MFMA variant of the GEMM engine for the adversarial step's wide convolutions (BASELINE configs[4] names bf16; the reference itself trains in fp32,
// so this is an OPT-IN precision of the discriminator: escx_disc_set_precision).  Out = epilogue( A_loader(m, k) . W[n][k] ), operands rounded to bf16
// (round to nearest even) when they are staged into LDS, products and sums in fp32 on v_mfma_f32_16x16x32_bf16 - 16x the matrix rate of the fp32 MFMA.
//
//  * same loader / epilogue interfaces as gemm_engine.h (the implicit-GEMM gathers ConvSU / ConvTS / ConvTSP and the dX / activation epilogues are shared):
//    the feature maps and the packed weights stay fp32 in HBM, every other kernel of the step is unchanged;
//  * same operand roles: the WEIGHT tile is the MFMA "A" operand, the ACTIVATION tile the "B" operand: lane (l15, lg) ends up with 4 consecutive output
//    features of one output row; a lane's 8 bf16 of an operand are 8 consecutive k of one tile row (one ds_read_b128 from the [row][k] LDS image) - the
//    k slots of the two operands pair up whatever the instruction's internal k order is;
//  * K step 32 (one MFMA deep), 128 x 128 workgroup tiles (256 x 128 as a tuning switch): at 16x the matrix rate the kernel is bound by what the L2 feeds
//    (fp32 operands: 32 FLOP per byte) and by the bytes in flight per CU, not by the matrix pipe;
//  * 1-D grid with the N tiles of one M tile adjacent (they gather the same activation rows at the same time), n tiles first in the block index.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"

namespace escx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (__bf16)v[e];           // v_cvt_pk_bf16_f32: round to nearest even
    return r;
}

// Three-term split of four fp32 values (NTERM = 3 instantiations, round 5): v = t0 + t1 + t2 exactly, each term bf16 (see fused_mlp_x3.h for the arithmetic:
// six exact cross products per operand pair, fp32 accumulation - fp32-grade results on the bf16 matrix cores).
__device__ __forceinline__ void split3_bf16x4(f32x4 v, bf16x4& t0, bf16x4& t1, bf16x4& t2) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 a1 = (__bf16)v[e];
        const float r1 = v[e] - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;
        t0[e] = a1; t1[e] = a2; t2[e] = (__bf16)r2;
    }
}

// WB16: the weight matrix is read from its bf16 image (the packed weights rounded once per parameter version: cvt_bf16_kernel below, disc.hip refresh_bf16_weights) - half the
// bytes of the weight operand and no conversion in the staging path; the rounding is the same nearest-even, so the results are bit-identical to WB16 = false.
// NTERM = 3: both operands are split into three bf16 terms while they are staged (three LDS planes per operand) and every tile product is the six leading cross terms,
// smallest first: the discriminator's "split" precision - fp32-grade arithmetic (escx_disc_set_precision(d, 2)), not the bf16 approximation of NTERM = 1.
template <int BM, int BN, bool WB16, class Loader, class Epi, int KS = 1, int NTERM = 1>       // KS: 32-deep MFMA steps per staged tile (one barrier pair per KS steps)
__global__ __launch_bounds__(256) void gemm_bf16_kernel(Loader ld, const float* __restrict__ Wt, const __bf16* __restrict__ Wt16, int M, int Np, int Kp, int nblk_n, Epi ep,
                                                        size_t w16_plane = 0) {     // NTERM = 3 with WB16: term p of the weights at Wt16 + p * w16_plane (split3_bf16_kernel)
    static_assert(BM % 64 == 0 && BN % 16 == 0, "tile shape");
    constexpr int BK = 32 * KS;
    constexpr int LD = BK + 8;                 // bf16 per LDS row: 80 B, the 16 rows of a fragment read start 20 banks apart
    constexpr int TM = BM / 64, TN = BN / 16;
    constexpr int KV = BK / 4;                 // float4 per tile row and K step
    constexpr int AJ = BM * KV / 256, BJ = BN * KV / 256;
    static_assert((BM * KV) % 256 == 0 && (BN * KV) % 256 == 0, "tile rows x K step must fill the workgroup");

    static_assert(NTERM == 1 || NTERM == 3, "one rounded term or three split terms");
    __shared__ __attribute__((aligned(16))) __bf16 As[NTERM * BM * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[NTERM * BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int bm = blockIdx.x / nblk_n, bn = blockIdx.x - bm * nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;

    typename Loader::Ctx ctx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) ctx[j] = ld.make_ctx(m0 + (tid + j * 256) / KV);

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = zero4();

    constexpr int P16 = BK / 8;                 // 16-byte pieces (8 bf16) per weight-tile row
    constexpr int BJ16 = BN * P16 / 256;        // ... per thread
    f32x4 ra[AJ], rb[WB16 ? 1 : BJ];
    uint4 rb16[WB16 ? NTERM * BJ16 : 1];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) ra[j] = ld.load4(ctx[j], k0, 4 * ((tid + j * 256) % KV));
        if constexpr (WB16) {
#pragma unroll
            for (int j = 0; j < BJ16; ++j) {
                const int i = tid + j * 256, row = i / P16, c8 = i % P16;
#pragma unroll
                for (int p = 0; p < NTERM; ++p)
                    rb16[p * BJ16 + j] = (n0 + row < Np && k0 + 8 * c8 < Kp) ? *reinterpret_cast<const uint4*>(Wt16 + p * w16_plane + (size_t)(n0 + row) * Kp + k0 + 8 * c8) : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int i = tid + j * 256, row = i / KV, c4 = i % KV;
                rb[j] = (n0 + row < Np && k0 + 4 * c4 < Kp) ? ld4(Wt + (size_t)(n0 + row) * Kp + k0 + 4 * c4) : zero4();       // K tail of a ragged last step: zero
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Kp; k0 += BK) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            if constexpr (NTERM == 3) {
                bf16x4 t0, t1, t2; split3_bf16x4(ra[j], t0, t1, t2);
                *reinterpret_cast<bf16x4*>(&As[(i / KV) * LD + 4 * (i % KV)]) = t0;
                *reinterpret_cast<bf16x4*>(&As[BM * LD + (i / KV) * LD + 4 * (i % KV)]) = t1;
                *reinterpret_cast<bf16x4*>(&As[2 * BM * LD + (i / KV) * LD + 4 * (i % KV)]) = t2;
            } else {
                *reinterpret_cast<bf16x4*>(&As[(i / KV) * LD + 4 * (i % KV)]) = to_bf16x4(ra[j]);
            }
        }
        if constexpr (WB16) {
#pragma unroll
            for (int j = 0; j < BJ16; ++j) {
                const int i = tid + j * 256;
#pragma unroll
                for (int p = 0; p < NTERM; ++p) *reinterpret_cast<uint4*>(&Bs[p * BN * LD + (i / P16) * LD + 8 * (i % P16)]) = rb16[p * BJ16 + j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int i = tid + j * 256;
                if constexpr (NTERM == 3) {
                    bf16x4 t0, t1, t2; split3_bf16x4(rb[j], t0, t1, t2);
                    *reinterpret_cast<bf16x4*>(&Bs[(i / KV) * LD + 4 * (i % KV)]) = t0;
                    *reinterpret_cast<bf16x4*>(&Bs[BN * LD + (i / KV) * LD + 4 * (i % KV)]) = t1;
                    *reinterpret_cast<bf16x4*>(&Bs[2 * BN * LD + (i / KV) * LD + 4 * (i % KV)]) = t2;
                } else {
                    *reinterpret_cast<bf16x4*>(&Bs[(i / KV) * LD + 4 * (i % KV)]) = to_bf16x4(rb[j]);
                }
            }
        }
        __syncthreads();
        if (k0 + BK < Kp) fetch(k0 + BK);
        if constexpr (NTERM == 3) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 af3[3][TM];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int b = 0; b < TM; ++b) af3[p][b] = *reinterpret_cast<const bf16x8*>(&As[p * BM * LD + (wave * (BM / 4) + b * 16 + l15) * LD + 32 * ks + 8 * lg]);
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    bf16x8 wf3[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) wf3[p] = *reinterpret_cast<const bf16x8*>(&Bs[p * BN * LD + (a * 16 + l15) * LD + 32 * ks + 8 * lg]);
#define ESCX_G3(I, J) _Pragma("unroll") for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf3[I], af3[J][b], acc[a][b], 0, 0, 0);
                    ESCX_G3(0, 2) ESCX_G3(2, 0) ESCX_G3(1, 1) ESCX_G3(0, 1) ESCX_G3(1, 0) ESCX_G3(0, 0)
#undef ESCX_G3
                }
            }
        } else
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) af[b] = *reinterpret_cast<const bf16x8*>(&As[(wave * (BM / 4) + b * 16 + l15) * LD + 32 * ks + 8 * lg]);
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&Bs[(a * 16 + l15) * LD + 32 * ks + 8 * lg]);
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[b], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wave * (BM / 4) + b * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + a * 16 + 4 * lg;
            if (n < Np) ep.store(m, n, acc[a][b], 0);
        }
    }
}

// Shapes this variant takes: K a multiple of the 32-wide step, outputs a multiple of the 128-wide tile, enough rows to fill the chip with the chosen tile.
inline bool bf16_gemm_ok(int M, int Np, int Kp) { return Np % 128 == 0 && Kp % 32 == 0 && Kp >= 256 && M >= 1024; }

static __global__ void split3_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n4, size_t plane) {      // n4 float4s -> three planes of bf16x4s
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        bf16x4 t0, t1, t2; split3_bf16x4(reinterpret_cast<const f32x4*>(src)[i], t0, t1, t2);
        reinterpret_cast<bf16x4*>(dst)[i] = t0; reinterpret_cast<bf16x4*>(dst + plane)[i] = t1; reinterpret_cast<bf16x4*>(dst + 2 * plane)[i] = t2;
    }
}

static __global__ void cvt_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n4) {      // n4 float4s -> bf16x4s
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<bf16x4*>(dst)[i] = to_bf16x4(reinterpret_cast<const f32x4*>(src)[i]);
}

template <class Loader, class Epi>
inline void launch_gemm_bf16(const Loader& ld, const float* Wt, const __bf16* Wt16, int M, int Np, int Kp, const Epi& ep, hipStream_t s, int nterm = 1, size_t w16_plane = 0) {
    if (nterm == 3) {           // split precision: three terms per operand; the weights from their pre-split image when there is one, else split while staged (same terms)
        const dim3 grid(((M + 127) / 128) * (Np / 128));
        if (Wt16) hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, true, Loader, Epi, 1, 3>), grid, dim3(256), 0, s, ld, Wt, Wt16, M, Np, Kp, Np / 128, ep, w16_plane);
        else hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, false, Loader, Epi, 1, 3>), grid, dim3(256), 0, s, ld, Wt, Wt16, M, Np, Kp, Np / 128, ep, (size_t)0);
        return;
    }
    static const int env_bm = [] { const char* e = ESCX_TUNE_ENV("ESCX_BF16_BM"); return e ? atoi(e) : 0; }();       // tuning aid: 128 / 256 rows per workgroup
    static const int env_ks = [] { const char* e = ESCX_TUNE_ENV("ESCX_BF16_KS"); return e ? atoi(e) : 1; }();        // tuning aid: 2 = 64-deep staged tiles
    const int nbn = Np / 128;
    // 128 rows: 156 registers, three workgroups per CU; 256 rows: 272 registers, ONE wave per SIMD - measured on the 1024 -> 1024 period layer: 409 against
    // 271 TFLOP/s forward, 384 against 252 dX (the K step is one memory round trip deep, so the bytes in flight per CU decide)
    const bool big = env_bm == 256;
    if (big) hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, false, Loader, Epi>), dim3(((M + 255) / 256) * nbn), dim3(256), 0, s, ld, Wt, Wt16, M, Np, Kp, nbn, ep);
    else if (Wt16 && env_ks == 2 && Kp % 64 == 0) hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, true, Loader, Epi, 2>), dim3(((M + 127) / 128) * nbn), dim3(256), 0, s, ld, Wt, Wt16, M, Np, Kp, nbn, ep);
    else if (Wt16) hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, true, Loader, Epi>), dim3(((M + 127) / 128) * nbn), dim3(256), 0, s, ld, Wt, Wt16, M, Np, Kp, nbn, ep);
    else hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, false, Loader, Epi>), dim3(((M + 127) / 128) * nbn), dim3(256), 0, s, ld, Wt, Wt16, M, Np, Kp, nbn, ep);
}

// Row-major operand with a column bound (the linear layers of the training step through the split-operand kernels: widths that are no multiple of the 32-deep
// K step or of the 128-wide dW tile read zeros behind the last column).
struct PlainG {
    const float* A; int lda; int M; int N;
    typedef const float* Ctx;
    __device__ __forceinline__ Ctx make_ctx(int m) const { return m < M ? A + (size_t)m * lda : nullptr; }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const { return (c && k0 + kin < N) ? ld4(c + k0 + kin) : zero4(); }
};

// out = A . Wt^T with split (3 x bf16) operands for any epilogue of the fp32 engine: tile = 128 or 64 rows x 128 or 96 columns, least column padding first.
template <class Loader, class Epi>
inline void launch_gemm_x3(const Loader& ld, const float* Wt, int M, int Np, int Kp, const Epi& ep, hipStream_t s) {
    const int pad128 = (Np + 127) / 128 * 128, pad96 = (Np + 95) / 96 * 96;
    const bool n128 = pad128 <= pad96;
    const int nbn = n128 ? pad128 / 128 : pad96 / 96;
    const bool big = (long long)((M + 127) / 128) * nbn >= 512;          // two resident workgroups per CU, twice over: else 64-row tiles
    const dim3 grid(((M + (big ? 127 : 63)) / (big ? 128 : 64)) * nbn);
#define ESCX_X3_LAUNCH(BM, BN) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, Loader, Epi, 1, 3>), grid, dim3(256), 0, s, ld, Wt, (const __bf16*)nullptr, M, Np, Kp, nbn, ep, (size_t)0)
    if (big && n128) ESCX_X3_LAUNCH(128, 128);
    else if (big) ESCX_X3_LAUNCH(128, 96);
    else if (n128) ESCX_X3_LAUNCH(64, 128);
    else ESCX_X3_LAUNCH(64, 96);
#undef ESCX_X3_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// dW[n][k] = sum_m A[m][n] * B[m][k] on the bf16 MFMA (A = upstream-gradient rows, B = the gathered input rows), db[n] = sum_m A[m][n] in fp32.
// The contraction runs over ROWS, so both operands have to reach the MFMA transposed (8 consecutive m per lane): a thread loads rows (m, m + 1) of its four
// columns, rounds, packs the two rows into one 32-bit word and writes [column][m] into LDS - lanes walk m fastest, so the 64 words of a write hit 64 banks,
// and the fragment reads are the same conflict-free 16-byte reads as in the forward kernel.  One workgroup = one 128 x 128 tile of dW and one slice of M,
// 2 x 2 waves of 64 x 64; the X side is the MFMA "A" operand so that a lane holds four consecutive k of one n (16-byte stores into part[slice][n][k]).
// Slices are added in increasing order by reduce_partials, as in the fp32 kernels.
// ------------------------------------------------------------------------------------------------
template <class LdA, class LdB, int NTERM = 1>       // NTERM = 3: three-term split of both operands, six cross terms per tile product (the "split" precision)
__global__ __launch_bounds__(256) void gemm_dw_bf16_kernel(LdA la, LdB lb, int M, int Np, int Kp, int nblk_k, int m_per_slice, float* __restrict__ part,
                                                           float* __restrict__ bpart) {
    constexpr int T = 128, MS = 32, LD = MS + 8;
    __shared__ __attribute__((aligned(16))) __bf16 At[NTERM * T * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bt[NTERM * T * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int mp = lane & 15, cq = 4 * (4 * wave + (lane >> 4));          // staging role: row pair mp of the step, columns cq .. cq + 3 (and + 64)
    const int bn = blockIdx.x / nblk_k, bk = blockIdx.x - bn * nblk_k;
    const int n0 = bn * T, k0 = bk * T;
    const int mbeg = blockIdx.y * m_per_slice, mend = min(M, mbeg + m_per_slice);
    const int wi = wave & 1, wj = wave >> 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero4();
    f32x4 bs[2] = {zero4(), zero4()};
    f32x4 ra[2][2], rb[2][2];                   // [column group][row of the pair]
    auto fetch = [&](int m0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = m0 + 2 * mp + r;
            const typename LdA::Ctx ca = la.make_ctx(m < mend ? m : M);         // rows behind the slice (or the matrix) read as zero
            const typename LdB::Ctx cb = lb.make_ctx(m < mend ? m : M);
#pragma unroll
            for (int j = 0; j < 2; ++j) { ra[j][r] = la.load4(ca, n0 + 64 * j + cq, 0); rb[j][r] = lb.load4(cb, k0 + 64 * j + cq, 0); }
        }
    };
    auto pack2 = [](float lo, float hi) -> unsigned {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 p; p[0] = (__bf16)lo; p[1] = (__bf16)hi;
        return __builtin_bit_cast(unsigned, p);
    };
    if (mbeg < mend) fetch(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += MS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bs[j] += ra[j][0] + ra[j][1];
            if constexpr (NTERM == 3) {
                bf16x4 a0[3], a1[3], b0[3], b1[3];              // [term] of rows m and m + 1
                split3_bf16x4(ra[j][0], a0[0], a0[1], a0[2]); split3_bf16x4(ra[j][1], a1[0], a1[1], a1[2]);
                split3_bf16x4(rb[j][0], b0[0], b0[1], b0[2]); split3_bf16x4(rb[j][1], b1[0], b1[1], b1[2]);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        bf16x2 pa; pa[0] = a0[p][e]; pa[1] = a1[p][e];
                        bf16x2 pb; pb[0] = b0[p][e]; pb[1] = b1[p][e];
                        *reinterpret_cast<unsigned*>(&At[p * T * LD + (64 * j + cq + e) * LD + 2 * mp]) = __builtin_bit_cast(unsigned, pa);
                        *reinterpret_cast<unsigned*>(&Bt[p * T * LD + (64 * j + cq + e) * LD + 2 * mp]) = __builtin_bit_cast(unsigned, pb);
                    }
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<unsigned*>(&At[(64 * j + cq + e) * LD + 2 * mp]) = pack2(ra[j][0][e], ra[j][1][e]);
                *reinterpret_cast<unsigned*>(&Bt[(64 * j + cq + e) * LD + 2 * mp]) = pack2(rb[j][0][e], rb[j][1][e]);
            }
            }
        }
        __syncthreads();
        if (m0 + MS < mend) fetch(m0 + MS);
        if constexpr (NTERM == 3) {
            bf16x8 af3[3][4];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int b = 0; b < 4; ++b) af3[p][b] = *reinterpret_cast<const bf16x8*>(&At[p * T * LD + (64 * wj + 16 * b + l15) * LD + 8 * lg]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x8 wf3[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) wf3[p] = *reinterpret_cast<const bf16x8*>(&Bt[p * T * LD + (64 * wi + 16 * a + l15) * LD + 8 * lg]);
#define ESCX_D3(I, J) _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf3[I], af3[J][b], acc[a][b], 0, 0, 0);
                ESCX_D3(0, 2) ESCX_D3(2, 0) ESCX_D3(1, 1) ESCX_D3(0, 1) ESCX_D3(1, 0) ESCX_D3(0, 0)
#undef ESCX_D3
            }
            __syncthreads();
            continue;
        }
        bf16x8 af[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) af[b] = *reinterpret_cast<const bf16x8*>(&At[(64 * wj + 16 * b + l15) * LD + 8 * lg]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&Bt[(64 * wi + 16 * a + l15) * LD + 8 * lg]);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    float* pt = part + (size_t)blockIdx.y * Np * Kp;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n0 + 64 * wj + 16 * b + l15;
        if (n >= Np) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int k = k0 + 64 * wi + 16 * a + 4 * lg;
            if (k < Kp) st4(pt + (size_t)n * Kp + k, acc[a][b]);
        }
    }
    if (bpart && bk == 0) {                     // column sums of the upstream rows this thread staged, over the 16 row pairs of a step: a fixed shuffle tree
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bs[j][e];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 16);
                bs[j][e] = v;
            }
            const int n = n0 + 64 * j + cq;
            if (mp == 0 && n < Np) st4(bpart + (size_t)blockIdx.y * Np + n, bs[j]);
        }
    }
}

// The same contraction for 32 output rows (the 32-channel band stacks): one workgroup = 32 x 128 tile of dW and one slice of M, wave w owns the k columns
// [32 w, 32 w + 32) x all 32 n; the upstream rows are staged by waves 0 and 1 (32 columns), the gathered input rows by everybody.
template <class LdA, class LdB>
__global__ __launch_bounds__(256) void gemm_dw_bf16_n32_kernel(LdA la, LdB lb, int M, int Np, int Kp, int nblk_k, int m_per_slice, float* __restrict__ part,
                                                               float* __restrict__ bpart) {
    constexpr int TB = 128, MS = 32, LD = MS + 8;
    __shared__ __attribute__((aligned(16))) __bf16 At[32 * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bt[TB * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int mp = lane & 15, cq = 4 * (4 * wave + (lane >> 4));
    const int k0 = blockIdx.x * TB;
    const int mbeg = blockIdx.y * m_per_slice, mend = min(M, mbeg + m_per_slice);
    const bool stage_a = wave < 2;              // columns cq .. cq + 3 < 32
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = zero4();
    f32x4 bs = zero4();
    f32x4 ra[2], rb[2][2];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = m0 + 2 * mp + r;
            const typename LdB::Ctx cb = lb.make_ctx(m < mend ? m : M);
#pragma unroll
            for (int j = 0; j < 2; ++j) rb[j][r] = lb.load4(cb, k0 + 64 * j + cq, 0);
            if (stage_a) { const typename LdA::Ctx ca = la.make_ctx(m < mend ? m : M); ra[r] = la.load4(ca, cq, 0); }
        }
    };
    auto pack2 = [](float lo, float hi) -> unsigned {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 p; p[0] = (__bf16)lo; p[1] = (__bf16)hi;
        return __builtin_bit_cast(unsigned, p);
    };
    if (mbeg < mend) fetch(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += MS) {
        if (stage_a) {
            bs += ra[0] + ra[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned*>(&At[(cq + e) * LD + 2 * mp]) = pack2(ra[0][e], ra[1][e]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned*>(&Bt[(64 * j + cq + e) * LD + 2 * mp]) = pack2(rb[j][0][e], rb[j][1][e]);
        __syncthreads();
        if (m0 + MS < mend) fetch(m0 + MS);
        bf16x8 af[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) af[b] = *reinterpret_cast<const bf16x8*>(&At[(16 * b + l15) * LD + 8 * lg]);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&Bt[(32 * wave + 16 * a + l15) * LD + 8 * lg]);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    float* pt = part + (size_t)blockIdx.y * Np * Kp;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = 16 * b + l15;
        if (n >= Np) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int k = k0 + 32 * wave + 16 * a + 4 * lg;
            if (k < Kp) st4(pt + (size_t)n * Kp + k, acc[a][b]);
        }
    }
    if (bpart && blockIdx.x == 0 && stage_a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = bs[e];
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 16);
            bs[e] = v;
        }
        if (mp == 0 && cq < Np) st4(bpart + (size_t)blockIdx.y * Np + cq, bs);
    }
}

}  // namespace escx
