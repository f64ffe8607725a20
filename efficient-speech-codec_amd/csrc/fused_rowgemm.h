// Fused LayerNorm + linear for the scale changes (PatchMerge: scale.py:97-115, PatchSplit: scale.py:131-145).
//
// Same register-resident scheme as the fused MLP: a wave gathers its 16*TM rows (1 or 2 source segments per row)
// straight into the MFMA operand layout, LayerNorms them in registers and multiplies by the weight matrix whose
// 16-row tiles stream through a double-buffered LDS ring (fragment order, global_load_lds).  Each accumulator tile is
// stored as soon as it is complete (16 B per lane) -- plain rows for PatchMerge, pixel-shuffled rows for PatchSplit.
// The normalised tensor never goes to memory (the unfused path writes and re-reads it).
//
// Grid = row blocks x OUTPUT-COLUMN chunks.  At the deep scales a 36-clip batch has fewer 16-row tiles than the chip has SIMDs
// (C = 384: 675 per half batch), and a wave that walks all 24 output tiles alone on its SIMD exposes every DMA wait and barrier
// (34 % MFMA-busy, 110 us for a 20 us contraction).  Splitting the OUTPUT columns over workgroups multiplies the waves without touching
// any reduction: every output element is still one k-ordered fmaf chain, so results are bit-identical for every chunking; the cost is
// the re-done row gather + LayerNorm (K loads per row per chunk), chosen on the host (launch_rowgemm).
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"

namespace escx {

struct RowGemmArgs {
    const float* x;             // [B*src_rows][Cp]
    float* out;
    const float* gamma; const float* beta;      // [SEGS][Cp]
    const f32x4* wf;            // [N tiles][KK][64] fragments
    const int* map;             // SEGS == 2: [rows_per_clip][2] source rows (or -1); SEGS == 1: unused
    int M, rows_per_clip, src_rows_per_clip, C, Cp, NT;     // NT = output tiles of 16
    int split, H, W, C2p;       // split != 0: out[(b, 2h+s, w)][c] with n = s*C2p + c ; else out[m][n], row stride 16*NT
    float eps;
    int nt_chunk;               // output tiles per workgroup column: blockIdx.y owns tiles [y * nt_chunk, (y + 1) * nt_chunk)
};

template <int KP, int SEGS, int TM, int NW, int UT>
__global__ __launch_bounds__(64 * NW) void rowgemm_fused_kernel(RowGemmArgs a) {
    constexpr int KK = KP / 16;
    __shared__ f32x4 wbuf[2][UT * KK * 64];
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: keeps the DMA issue loop scalar
    const int m0 = (blockIdx.x * NW + wave) * (16 * TM);
    const int nt_lo = blockIdx.y * a.nt_chunk, nt_hi = min(a.NT, nt_lo + a.nt_chunk);
    const int n_stages = (nt_hi - nt_lo + UT - 1) / UT;

    auto issue = [&](int st, int buf) {
        const int cnt = min(UT, nt_hi - nt_lo - st * UT) * KK;
        const f32x4* src = a.wf + (size_t)(nt_lo + st * UT) * KK * 64 + lane;
        for (int c = wave; c < cnt; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&wbuf[buf][c * 64]), 16, 0, 0);
    };
    issue(0, 0);

    const int SEGK = KP / SEGS;                 // = Cp
    f32x4 xf[TM][KK];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = m0 + t * 16 + l15;
        const float* sp[SEGS];
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
            sp[s] = nullptr;
            if (row < a.M) {
                const int b = row / a.rows_per_clip, rr = row - b * a.rows_per_clip;
                const int srow = (SEGS == 1) ? rr : a.map[rr * SEGS + s];
                if (srow >= 0) sp[s] = a.x + ((size_t)b * a.src_rows_per_clip + srow) * a.Cp;
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int k = 16 * kk + 4 * lg;
            const int s = k / SEGK, c = k - s * SEGK;
            xf[t][kk] = sp[s] ? ld4(sp[s] + c) : zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += xf[t][kk][e];          // pad channels are exact zeros
        }
        sum = sum_groups(sum);
        const float mean = sum / (float)(SEGS * a.C);
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int k = 16 * kk + 4 * lg; const int c = k % SEGK;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xf[t][kk][e] - mean; v += d * d; }
        }
        v = sum_groups(v) - (float)(SEGS * (SEGK - a.C)) * mean * mean;     // the zero pads each added mean^2
        const float rstd = 1.0f / sqrtf(v / (float)(SEGS * a.C) + a.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int k = 16 * kk + 4 * lg; const int c = k % SEGK;
            const f32x4 g = ld4(a.gamma + k), bb = ld4(a.beta + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) xf[t][kk][e] = (xf[t][kk][e] - mean) * rstd * g[e] + bb[e];       // gamma = beta = 0 in the pads
        }
    }

    // output row bases (pixel shuffle for split)
    size_t obase[TM][2];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = m0 + t * 16 + l15;
        if (a.split) {
            const int b = row / (a.H * a.W); const int r0 = row - b * a.H * a.W; const int h = r0 / a.W, w = r0 - h * a.W;
            obase[t][0] = ((size_t)(b * 2 * a.H + 2 * h) * a.W + w) * a.C2p;
            obase[t][1] = ((size_t)(b * 2 * a.H + 2 * h + 1) * a.W + w) * a.C2p;
        } else {
            obase[t][0] = obase[t][1] = (size_t)row * (16 * a.NT);
        }
    }

    for (int st = 0; st < n_stages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < n_stages) issue(st + 1, (st + 1) & 1);
        const f32x4* wb = &wbuf[st & 1][lane];
        const int nt_end = min(nt_hi, nt_lo + (st + 1) * UT);
        for (int nt = nt_lo + st * UT; nt < nt_end; ++nt, wb += KK * 64) {
            f32x4 acc[TM], acc2[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) { acc[t] = zero4(); acc2[t] = zero4(); }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const f32x4 w = wb[kk * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        if (TM == 1 && (r & 1)) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], acc2[t], 0, 0, 0);
                        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], acc[t], 0, 0, 0);
                    }
            }
            const int n = 16 * nt + 4 * lg;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (m0 + t * 16 + l15 >= a.M) continue;
                const f32x4 v = TM == 1 ? acc[t] + acc2[t] : acc[t];
                if (a.split) { const int s = n / a.C2p; st4(a.out + obase[t][s] + (n - s * a.C2p), v); }
                else st4(a.out + obase[t][0] + n, v);
            }
        }
    }
}

}  // namespace escx
