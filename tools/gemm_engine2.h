// Double-buffered variant of the fp32 MFMA GEMM engine (gemm_engine.h) for the MFMA-bound shapes of the training / adversarial step (the
// discriminator's 512 / 1024-channel convolutions, the C >= 96 linear layers): same loaders, same epilogues, same fragment layout and the SAME
// arithmetic (every output element is one k-ordered fmaf chain: results are bitwise those of gemm_kernel), but
//   * the LDS tile is double-buffered: the next K step's tile is written into the other buffer AFTER the current step's MFMAs were issued, so
//     the global-load wait sits behind the wave's own matrix work and there is ONE barrier per K step instead of two;
//   * wider workgroup tiles are available (BM = 256: 4 x 6 accumulator tiles per wave), which halve the weight re-reads per output row.
#pragma once
#include "gemm_engine.h"

namespace escx {

template <int BM, int BN, int BK, class Loader, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel2(Loader ld, const float* __restrict__ Wt, int M, int Np, int Kp, int k_per_z, Epi ep) {
    static_assert(BM % 64 == 0 && BN % 16 == 0 && BK % 16 == 0, "tile shape");
    constexpr int LDS_LD = BK + 4;
    constexpr int TM = BM / 64;
    constexpr int TN = BN / 16;
    constexpr int KV = BK / 4;
    constexpr int A4 = BM * KV;
    constexpr int B4 = BN * KV;
    constexpr int AJ = (A4 + 255) / 256;
    constexpr int BJ = (B4 + 255) / 256;

    __shared__ float As[2][BM * LDS_LD];
    __shared__ float Bs[2][BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * k_per_z;
    const int kend = min(Kp, kbeg + k_per_z);

    typename Loader::Ctx ctx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int i = tid + j * 256;
        ctx[j] = ld.make_ctx(m0 + (i < A4 ? i / KV : 0));
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = zero4();

    f32x4 ra[AJ], rb[BJ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            ra[j] = (A4 % 256 == 0 || i < A4) ? ld.load4(ctx[j], k0, 4 * (i % KV)) : zero4();
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256;
            const int row = i / KV, c4 = i % KV;
            rb[j] = ((B4 % 256 == 0 || i < B4) && n0 + row < Np) ? ld4(Wt + (size_t)(n0 + row) * Kp + k0 + 4 * c4) : zero4();
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            if (A4 % 256 == 0 || i < A4) st4(&As[buf][(i / KV) * LDS_LD + 4 * (i % KV)], ra[j]);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256;
            if (B4 % 256 == 0 || i < B4) st4(&Bs[buf][(i / KV) * LDS_LD + 4 * (i % KV)], rb[j]);
        }
    };
    if (kbeg < kend) { fetch(kbeg); stage(0); }
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, cur ^= 1) {
        const bool more = k0 + BK < kend;
        if (more) fetch(k0 + BK);
        const float* as = &As[cur][(wave * (BM / 4) + l15) * LDS_LD + 4 * lg];
        const float* bs = &Bs[cur][l15 * LDS_LD + 4 * lg];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b) af[b] = ld4(as + b * 16 * LDS_LD + kk);
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[a] = ld4(bs + a * 16 * LDS_LD + kk);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a][r], af[b][r], acc[a][b], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
    }
    if constexpr (epi_is_rowwise<Epi>::value) {
        ep.template finish<TN, TM>(acc, m0 + wave * (BM / 4), lane, M);
        return;
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wave * (BM / 4) + b * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + a * 16 + 4 * lg;
            if (n < Np) ep.store(m, n, acc[a][b], blockIdx.z);
        }
    }
}

template <int BM, int BN, int BK, class Loader, class Epi>
inline void launch_gemm2(const Loader& ld, const float* Wt, int M, int Np, int Kp, const Epi& ep, hipStream_t s, int splits = 1) {
    int kIters = Kp / BK;
    int per = (kIters + splits - 1) / splits;
    dim3 grid((M + BM - 1) / BM, (Np + BN - 1) / BN, (kIters + per - 1) / per);
    hipLaunchKernelGGL((gemm_kernel2<BM, BN, BK, Loader, Epi>), grid, dim3(256), 0, s, ld, Wt, M, Np, Kp, per * BK, ep);
}

}  // namespace escx

namespace escx {

// Software-pipelined form: K steps of 16 through a THREE-slot LDS ring and two register sets of MFMA fragments.
// In iteration t a wave (1) writes tile t+2 into the ring (its global loads were issued a whole iteration earlier), (2) issues the global loads of tile
// t+3, (3) issues the LDS reads of tile t+1's fragments into the idle register set, and only then (4) runs the 12 x 4 MFMAs of tile t, whose fragments
// were read an iteration ago - so neither an LDS read, nor an LDS write, nor a global load is ever waited for with an empty matrix pipe; the one barrier
// per step comes right after the MFMA batch and costs the skew between the four waves only.  Measured against gemm_kernel on the discriminator's
// 1024 x 5120 contraction (tools/ubench_gemm.hip): see DESIGN.md section 7c.  Row stride BK + 8 dwords: the 16-lane groups of a ds_read_b128 then
// touch 16 distinct bank quads (stride BK + 4 serialises three of them: SQ_LDS_BANK_CONFLICT = 50 % of the LDS-array cycles of gemm_kernel).
// Arithmetic unchanged: every output element is the same k-ordered fmaf chain as in gemm_kernel (bitwise identical results).
template <int BM, int BN, class Loader, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel3(Loader ld, const float* __restrict__ Wt, int M, int Np, int Kp, int k_per_z, Epi ep) {
    constexpr int BK = 16;
    constexpr int LDS_LD = BK + 8;
    constexpr int TM = BM / 64;
    constexpr int TN = BN / 16;
    constexpr int KV = BK / 4;
    constexpr int A4 = BM * KV;
    constexpr int B4 = BN * KV;
    constexpr int AJ = (A4 + 255) / 256;
    constexpr int BJ = (B4 + 255) / 256;
    static_assert(BM % 64 == 0 && BN % 16 == 0, "tile shape");

    __shared__ float As[3][BM * LDS_LD];
    __shared__ float Bs[3][BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * k_per_z;
    const int kend = min(Kp, kbeg + k_per_z);
    const int nk = (kend - kbeg + BK - 1) / BK;

    typename Loader::Ctx ctx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int i = tid + j * 256;
        ctx[j] = ld.make_ctx(m0 + (i < A4 ? i / KV : 0));
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = zero4();

    f32x4 ra[AJ], rb[BJ];
    auto fetch = [&](int t) {
        const int k0 = kbeg + t * BK;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            ra[j] = (A4 % 256 == 0 || i < A4) ? ld.load4(ctx[j], k0, 4 * (i % KV)) : zero4();
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256;
            const int row = i / KV, c4 = i % KV;
            rb[j] = ((B4 % 256 == 0 || i < B4) && n0 + row < Np) ? ld4(Wt + (size_t)(n0 + row) * Kp + k0 + 4 * c4) : zero4();
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            if (A4 % 256 == 0 || i < A4) st4(&As[buf][(i / KV) * LDS_LD + 4 * (i % KV)], ra[j]);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256;
            if (B4 % 256 == 0 || i < B4) st4(&Bs[buf][(i / KV) * LDS_LD + 4 * (i % KV)], rb[j]);
        }
    };
    f32x4 af[2][TM], wf[2][TN];
    auto frags = [&](int set, int buf) {
        const float* as = &As[buf][(wave * (BM / 4) + l15) * LDS_LD + 4 * lg];
        const float* bs = &Bs[buf][l15 * LDS_LD + 4 * lg];
#pragma unroll
        for (int b = 0; b < TM; ++b) af[set][b] = ld4(as + b * 16 * LDS_LD);
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[set][a] = ld4(bs + a * 16 * LDS_LD);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[set][a][r], af[set][b][r], acc[a][b], 0, 0, 0);
    };

    if (nk > 0) {
        fetch(0); stage(0);
        if (nk > 1) fetch(1);
        __syncthreads();
        frags(0, 0);
        if (nk > 1) stage(1);
        if (nk > 2) fetch(2);
        __syncthreads();
    }
    // two iterations per trip so that the fragment register sets are compile-time indices
    int b0 = 0;                                  // ring slot of tile t
    for (int t = 0; t < nk; t += 2) {
        {
            const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            if (t + 2 < nk) stage(b2);
            if (t + 3 < nk) fetch(t + 3);
            if (t + 1 < nk) frags(1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            b0 = b1;
        }
        if (t + 1 < nk) {
            const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            if (t + 3 < nk) stage(b2);
            if (t + 4 < nk) fetch(t + 4);
            if (t + 2 < nk) frags(0, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            b0 = b1;
        }
    }
    if constexpr (epi_is_rowwise<Epi>::value) {
        ep.template finish<TN, TM>(acc, m0 + wave * (BM / 4), lane, M);
        return;
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wave * (BM / 4) + b * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + a * 16 + 4 * lg;
            if (n < Np) ep.store(m, n, acc[a][b], blockIdx.z);
        }
    }
}

template <int BM, int BN, class Loader, class Epi>
inline void launch_gemm3(const Loader& ld, const float* Wt, int M, int Np, int Kp, const Epi& ep, hipStream_t s, int splits = 1) {
    constexpr int BK = 16;
    int kIters = Kp / BK;
    int per = (kIters + splits - 1) / splits;
    dim3 grid((M + BM - 1) / BM, (Np + BN - 1) / BN, (kIters + per - 1) / per);
    hipLaunchKernelGGL((gemm_kernel3<BM, BN, Loader, Epi>), grid, dim3(256), 0, s, ld, Wt, M, Np, Kp, per * BK, ep);
}

}  // namespace escx

namespace escx {

// Round 5: the engine's loop on v_mfma_f32_32x32x2_f32 instead of v_mfma_f32_16x16x4_f32.  Same loaders, epilogues, LDS image
// ([row][k], stride BK + 4) and staging as gemm_kernel; a wave owns 32 rows (one column tile of the B operand) x BN / 32 weight tiles, its 16
// accumulator values per tile are D[n = 8 q + 4 (lane / 32) + r][m = lane % 32]: still 4 consecutive output features per f32x4 store.
// Within a 16-wide k chunk, MFMA step s takes k = 8 * (lane / 32) + s (each lane reads 8 consecutive floats = two ds_read_b128), so the
// SUMMATION ORDER differs from the engine's - tolerance-level agreement only (loss terms, not codes).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int BK, class Loader, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel_m32(Loader ld, const float* __restrict__ Wt, int M, int Np, int Kp, int k_per_z, Epi ep) {
    static_assert(BM == 128 && BN % 32 == 0 && BK % 16 == 0, "tile shape");
    constexpr int LDS_LD = BK + 4;
    constexpr int TN = BN / 32;
    constexpr int KV = BK / 4;
    constexpr int A4 = BM * KV, B4 = BN * KV;
    constexpr int AJ = (A4 + 255) / 256, BJ = (B4 + 255) / 256;
    __shared__ float As[BM * LDS_LD];
    __shared__ float Bs[BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hf = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * k_per_z, kend = min(Kp, kbeg + k_per_z);
    typename Loader::Ctx ctx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) { const int i = tid + j * 256; ctx[j] = ld.make_ctx(m0 + (i < A4 ? i / KV : 0)); }
    f32x16 acc[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    f32x4 ra[AJ], rb[BJ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) { const int i = tid + j * 256; ra[j] = (A4 % 256 == 0 || i < A4) ? ld.load4(ctx[j], k0, 4 * (i % KV)) : zero4(); }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256; const int row = i / KV, c4 = i % KV;
            rb[j] = ((B4 % 256 == 0 || i < B4) && n0 + row < Np) ? ld4(Wt + (size_t)(n0 + row) * Kp + k0 + 4 * c4) : zero4();
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) { const int i = tid + j * 256; if (A4 % 256 == 0 || i < A4) st4(&As[(i / KV) * LDS_LD + 4 * (i % KV)], ra[j]); }
#pragma unroll
        for (int j = 0; j < BJ; ++j) { const int i = tid + j * 256; if (B4 % 256 == 0 || i < B4) st4(&Bs[(i / KV) * LDS_LD + 4 * (i % KV)], rb[j]); }
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            f32x4 af[2], wf[TN][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) af[q] = ld4(&As[(wave * 32 + l31) * LDS_LD + kk + 8 * hf + 4 * q]);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int q = 0; q < 2; ++q) wf[a][q] = ld4(&Bs[(a * 32 + l31) * LDS_LD + kk + 8 * hf + 4 * q]);
#pragma unroll
            for (int st = 0; st < 8; ++st)
#pragma unroll
                for (int a = 0; a < TN; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[a][st >> 2][st & 3], af[st >> 2][st & 3], acc[a], 0, 0, 0);
        }
        __syncthreads();
    }
    const int m = m0 + wave * 32 + l31;
    if (m >= M) return;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + a * 32 + 8 * q + 4 * hf;
            if (n < Np) ep.store(m, n, f32x4{acc[a][4 * q], acc[a][4 * q + 1], acc[a][4 * q + 2], acc[a][4 * q + 3]}, blockIdx.z);
        }
}

template <int BN, int BK, class Loader, class Epi>
inline void launch_gemm_m32(const Loader& ld, const float* Wt, int M, int Np, int Kp, const Epi& ep, hipStream_t s) {
    dim3 grid((M + 127) / 128, (Np + BN - 1) / BN, 1);
    hipLaunchKernelGGL((gemm_kernel_m32<128, BN, BK, Loader, Epi>), grid, dim3(256), 0, s, ld, Wt, M, Np, Kp, Kp, ep);
}

}  // namespace escx
