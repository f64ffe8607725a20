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
#include "fused_attn.h"       // attn_split3, attn_x3_pack_kernel, bf16x8

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
    // combine-on-load (COMB instantiations): x is still split over the comb_n fc2 slabs of the last block's hidden-split MLP (see AttnArgs in fused_attn.h)
    const float* comb_partial; const float* comb_bias; long long comb_stride; int comb_n;
    const void* x3_wf;          // rowgemm_x3_kernel: split weight stream [output tile][3 KS fragments] (attn_x3_pack_kernel), else unused
    const float* x3_scale;      // NT = 2: {2^-k / sx, 2^k sx, sx} of the scaled two-term stream (its last 16 bytes)
};

template <int KP, int SEGS, int TM, int NW, int UT, bool COMB = false>
__global__ __launch_bounds__(64 * NW) void rowgemm_fused_kernel(RowGemmArgs a) {
    ESCX_SET_PRIO_SMALL();
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
            if constexpr (COMB) {
                f32x4 x = zero4();
                if (sp[s]) {
                    const size_t elem = (size_t)(sp[s] - a.x) + c;
                    f32x4 v = ld4(a.comb_partial + elem);
                    for (int h2 = 1; h2 < a.comb_n; ++h2) v += ld4(a.comb_partial + (size_t)h2 * a.comb_stride + elem);
                    v += ld4(a.comb_bias + c);
                    x = ld4(sp[s] + c) + v;
                }
                xf[t][kk] = x;
            } else {
                xf[t][kk] = sp[s] ? ld4(sp[s] + c) : zero4();
            }
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

// ------------------------------------------------------------------------------------------------
// Split-operand form (round 5; arithmetic: fused_mlp_x3.h): the K = SEGS * Cp contraction of PatchMerge / PatchSplit on v_mfma_f32_16x16x32_bf16 with
// the normalised rows and the weights each split exactly into three bf16 terms (six cross products, fp32 accumulation).  Same gather, same
// LayerNorm, same stores as rowgemm_fused_kernel; the weight stream is [output tile][3 KS fragments] (attn_x3_pack_kernel with no projection tiles,
// built from Layer::sub_wf), UT tiles per LDS stage.
// ------------------------------------------------------------------------------------------------
template <int KP, int SEGS, int TM, int NW, int UT, int NT = 3>      // NT = 2: two fp16 terms, three cross products (split_terms.h); the stream keeps its 3 KS fragment slots per tile
__global__ __launch_bounds__(64 * NW) void rowgemm_x3_kernel(RowGemmArgs a) {
    ESCX_SET_PRIO_SMALL();
    constexpr int KS = (KP + 31) / 32, TF = 3 * KS, SEGK = KP / SEGS;
    extern __shared__ __attribute__((aligned(16))) bf16x8 rg_wbuf[];            // [2][UT * TF * 64]
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = (blockIdx.x * NW + wave) * (16 * TM);
    const int n_stages = (a.NT + UT - 1) / UT;
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.x3_wf);

    auto issue = [&](int st, int buf) {
        const int cnt = min(UT, a.NT - st * UT) * TF;
        const bf16x8* src = wsrc + (size_t)(st * UT) * TF * 64 + lane;
        for (int c = wave; c < cnt; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&rg_wbuf[(buf * UT * TF + c) * 64]), 16, 0, 0);
    };
    issue(0, 0);

    bf16x8 xs[TM][NT][KS];
    float x2_dn = 1.f, x2_sx = 1.f;             // NT = 2: 2^-k / sx of the scaled weights, sx = the power-of-two scale of the LayerNorm output (split_terms.h range rule)
    if constexpr (NT == 2) { x2_dn = a.x3_scale[0]; x2_sx = a.x3_scale[2]; }
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
        float xv[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 32 * ks + 8 * lg;                     // 8 consecutive k of one segment (SEGK % 16 == 0)
            const int sg = k / SEGK, c = k - sg * SEGK;
            f32x4 v0 = zero4(), v1 = zero4();
            if (k < KP && sp[sg < SEGS ? sg : 0] && sg < SEGS) { v0 = ld4(sp[sg] + c); v1 = ld4(sp[sg] + c + 4); }     // rows past M, missing merge sources and the K padding read as zeros
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[ks][e] = v0[e]; xv[ks][4 + e] = v1[e]; sum += v0[e]; sum += v1[e]; }
        }
        sum = sum_groups(sum);
        const float mean = sum / (float)(SEGS * a.C);
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xv[ks][e] - mean; v += d * d; }
        v = sum_groups(v) - (float)(32 * KS - SEGS * a.C) * mean * mean;       // every zero slot added mean^2 (also the all-zero row of a missing merge source: as in the fp32 kernel)
        const float rstd = 1.0f / sqrtf(v / (float)(SEGS * a.C) + a.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 32 * ks + 8 * lg;
            f32x4 g0 = zero4(), g1 = zero4(), b0 = zero4(), b1 = zero4();
            if (k < KP) { g0 = ld4(a.gamma + k); g1 = ld4(a.gamma + k + 4); b0 = ld4(a.beta + k); b1 = ld4(a.beta + k + 4); }
            float xn[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xn[e] = (xv[ks][e] - mean) * rstd * g0[e] + b0[e];             // gamma = beta = 0 in the pads
                xn[4 + e] = (xv[ks][4 + e] - mean) * rstd * g1[e] + b1[e];
            }
            if constexpr (NT == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xn[e] *= x2_sx;
            }
            bf16x8 tt[NT];
            split_terms<NT>(xn, tt);
#pragma unroll
            for (int i = 0; i < NT; ++i) xs[t][i][ks] = tt[i];
        }
    }

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
        const bf16x8* wb = &rg_wbuf[((st & 1) * UT * TF) * 64 + lane];
        const int nt_end = min(a.NT, (st + 1) * UT);
        for (int nt = st * UT; nt < nt_end; ++nt, wb += TF * 64) {
            f32x4 acc[TM], acc2[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) { acc[t] = zero4(); acc2[t] = zero4(); }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 w[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) w[i] = wb[(ks * NT + i) * 64];
#define ESCX_RG_X3(I, J, D) _Pragma("unroll") for (int t = 0; t < TM; ++t) D[t] = mma_x<NT>(w[I], xs[t][J][ks], D[t]);
                if constexpr (NT == 3) { ESCX_RG_X3(0, 2, acc) ESCX_RG_X3(2, 0, acc2) ESCX_RG_X3(1, 1, acc) ESCX_RG_X3(0, 1, acc2) ESCX_RG_X3(1, 0, acc) ESCX_RG_X3(0, 0, acc2) }
                else { ESCX_RG_X3(0, 1, acc) ESCX_RG_X3(1, 0, acc2) ESCX_RG_X3(0, 0, acc) }
#undef ESCX_RG_X3
            }
            const int n = 16 * nt + 4 * lg;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (m0 + t * 16 + l15 >= a.M) continue;
                f32x4 v = acc[t] + acc2[t];
                if constexpr (NT == 2) v *= x2_dn;
                if (a.split) { const int s2 = n / a.C2p; st4(a.out + obase[t][s2] + (n - s2 * a.C2p), v); }
                else st4(a.out + obase[t][0] + n, v);
            }
        }
    }
}

#ifdef ESCX_EXPERIMENTAL       // tagged builds only (tune_env.h): both forms are faster alone and slower in the two-stream step, profiles/r4_rowgemm_ab.txt
// ------------------------------------------------------------------------------------------------
// Round 4 forms.  The streaming kernel above re-streams the whole weight matrix through LDS for every 16*TM*NW rows, issues each stage's
// DMA as one burst behind the barrier (~190 cycles per 1 KiB piece when a wave issues them back to back, tools/ubench_dma_cost.hip)
// and at the deep scales runs one wave per SIMD (256-320 registers): 34-45 % MFMA-busy, 40-58 us per launch for 15-20 us of work.
// Shared pieces: one row tile's gather and LayerNorm in the MFMA operand layout.
// ------------------------------------------------------------------------------------------------
template <int KP, int SEGS>
__device__ __forceinline__ void rowgemm_gather(const RowGemmArgs& a, int row0, int l15, int lg, f32x4 (&raw)[KP / 16]) {
    constexpr int KK = KP / 16, SEGK = KP / SEGS;
    const int row = row0 + l15;
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
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int k = 16 * kk + 4 * lg;
        const int s = k / SEGK, c = k - s * SEGK;
        raw[kk] = sp[s] ? ld4(sp[s] + c) : zero4();        // rows past M and missing merge sources read as zeros
    }
}
// Contraction is pinned (explicit fmaf, contract off): left to -ffp-contract=fast, hipcc fuses SOME of the d * d products of the variance
// into v_pk_fma and leaves others as mul + add, differently per instantiation - two kernels built from one source would round differently.
template <int KP, int SEGS>
__device__ __forceinline__ void rowgemm_normalise(const RowGemmArgs& a, int lg, f32x4 (&xf)[KP / 16]) {
#pragma clang fp contract(off)
    constexpr int KK = KP / 16, SEGK = KP / SEGS;
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += xf[kk][e];                 // pad channels are exact zeros
    sum = sum_groups(sum);
    const float mean = sum / (float)(SEGS * a.C);
    float v = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xf[kk][e] - mean; v = __builtin_fmaf(d, d, v); }
    v = __builtin_fmaf(-(float)(SEGS * (SEGK - a.C)) * mean, mean, sum_groups(v));       // the zero pads each added mean^2
    const float rstd = 1.0f / sqrtf(v / (float)(SEGS * a.C) + a.eps);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int k = 16 * kk + 4 * lg;
        const f32x4 gm = ld4(a.gamma + k), bb = ld4(a.beta + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[kk][e] = __builtin_fmaf((xf[kk][e] - mean) * rstd, gm[e], bb[e]);    // gamma = beta = 0 in the pads
    }
}
// element offset of output row `row` (segment s of a pixel-shuffled PatchSplit row)
__device__ __forceinline__ void rowgemm_obase(const RowGemmArgs& a, int row, size_t (&ob)[2]) {
    if (a.split) {
        const int b = row / (a.H * a.W); const int r0 = row - b * a.H * a.W; const int h = r0 / a.W, w = r0 - h * a.W;
        ob[0] = ((size_t)(b * 2 * a.H + 2 * h) * a.W + w) * a.C2p;
        ob[1] = ((size_t)(b * 2 * a.H + 2 * h + 1) * a.W + w) * a.C2p;
    } else {
        ob[0] = ob[1] = (size_t)row * (16 * a.NT);
    }
}
__device__ __forceinline__ void rowgemm_store(const RowGemmArgs& a, const size_t (&ob)[2], int n, f32x4 v) {
    if (a.split) { const int s = n / a.C2p; st4(a.out + ob[s] + (n - s * a.C2p), v); }
    else st4(a.out + ob[0] + n, v);
}

// ------------------------------------------------------------------------------------------------
// (A) Weight-stationary form for the SHALLOW scales (many rows, small matrices).  The workgroup loads ITS chunk of output tiles
// (<= ~144 KB of fragments) into LDS ONCE, and its waves then walk row tiles persistently: the steady state has no DMA, no barrier and no
// cross-wave dependency - a wave gathers + LayerNorms 16*TM rows in registers (optionally one group ahead), runs the tile GEMMs with one
// conflict-free ds_read_b128 per 4*TM MFMAs and stores each finished tile.  Matrices above the LDS budget are split over OUTPUT-COLUMN
// chunks (grid.y); a chunk's workgroups re-gather the rows.  Accumulator split as in rowgemm_fused_kernel (TM = 1: two chains).
// ------------------------------------------------------------------------------------------------
template <int KK> constexpr int rowgemm_ring() { return KK % 3 == 0 ? 3 : (KK % 5 == 0 ? 5 : (KK % 2 == 0 ? 2 : 1)); }

template <int KP, int SEGS, int TM, int NW, int WPS, bool PF>
__global__ __launch_bounds__(64 * NW, WPS) void rowgemm_ws_kernel(RowGemmArgs a) {
    constexpr int KK = KP / 16;
    constexpr int PD = rowgemm_ring<KK>();
    static_assert(KK % PD == 0, "the fragment ring must close on a tile boundary");
    extern __shared__ f32x4 wl[];               // [tiles of this chunk][KK][64] fragments
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt_lo = blockIdx.y * a.nt_chunk, nt_hi = min(a.NT, nt_lo + a.nt_chunk), ntc = nt_hi - nt_lo;
    {
        const f32x4* src = a.wf + (size_t)nt_lo * KK * 64 + lane;
        for (int c = wave; c < ntc * KK; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(wl + c * 64), 16, 0, 0);
    }
    const int n_groups = (a.M + 16 * TM - 1) / (16 * TM);
    const int stride = gridDim.x * NW;
    int g = blockIdx.x * NW + wave;

    f32x4 xf[TM][KK];
    if (g < n_groups) {
#pragma unroll
        for (int t = 0; t < TM; ++t) rowgemm_gather<KP, SEGS>(a, g * (16 * TM) + t * 16, l15, lg, xf[t]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of the chunk (and its first rows) have landed
    __syncthreads();                                    // ... and everybody else's: the only barrier of the kernel
    const f32x4* wbase = wl + lane;
    while (g < n_groups) {
#pragma unroll
        for (int t = 0; t < TM; ++t) rowgemm_normalise<KP, SEGS>(a, lg, xf[t]);
        const int gn = g + stride;
        f32x4 nxt[PF ? TM : 1][KK];
        if constexpr (PF) {                             // next group's rows fly under this group's MFMAs
            if (gn < n_groups) {
#pragma unroll
                for (int t = 0; t < TM; ++t) rowgemm_gather<KP, SEGS>(a, gn * (16 * TM) + t * 16, l15, lg, nxt[t]);
            }
        }
        size_t obase[TM][2];
#pragma unroll
        for (int t = 0; t < TM; ++t) rowgemm_obase(a, g * (16 * TM) + t * 16 + l15, obase[t]);
        f32x4 ring[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) ring[i] = wbase[i * 64];
        for (int j = 0; j < ntc; ++j) {
            const f32x4* wb = wbase + (size_t)j * KK * 64;
            const f32x4* wn = wbase + (size_t)min(j + 1, ntc - 1) * KK * 64;      // ring refill across the tile boundary (last tile: re-reads itself)
            f32x4 acc[TM], acc2[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) { acc[t] = zero4(); acc2[t] = zero4(); }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const f32x4 w = ring[kk % PD];
                ring[kk % PD] = (kk + PD < KK) ? wb[(kk + PD) * 64] : wn[(kk + PD - KK) * 64];
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        if (TM == 1 && (r & 1)) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], acc2[t], 0, 0, 0);
                        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], acc[t], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM, 0);
            }
            const int n = 16 * (nt_lo + j) + 4 * lg;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (g * (16 * TM) + t * 16 + l15 >= a.M) continue;
                rowgemm_store(a, obase[t], n, TM == 1 ? acc[t] + acc2[t] : acc[t]);
            }
        }
        g = gn;
        if constexpr (PF) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) xf[t][kk] = nxt[t][kk];
        } else {
            if (g < n_groups) {
#pragma unroll
                for (int t = 0; t < TM; ++t) rowgemm_gather<KP, SEGS>(a, g * (16 * TM) + t * 16, l15, lg, xf[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// (B) Shared-rows form for the DEEP scales (few rows, matrices of 100-600 KB).  A workgroup owns R row tiles: waves 0..R-1 gather +
// LayerNorm one tile each and park it in LDS in operand layout; after ONE barrier the NW waves split the OUTPUT tiles (NTW each) and
// every wave walks K once: per k-step R conflict-free ds_read_b128 (the rows) and NTW weight fragments fetched STRAIGHT from L2 into a
// register ring - no LDS staging, no DMA issue, no stage barriers (a wave's fragments are nobody else's) - feeding R * NTW * 4 MFMAs.
// Each fragment is used R times from registers, each row fragment NTW times.  CH = 2: the (r even | r odd) accumulator pair of the
// streaming kernel's TM = 1 instantiations (K > 192), CH = 1: its single chain (K <= 192) - every output element is the same k-ordered sum.
// ------------------------------------------------------------------------------------------------
template <int KP, int SEGS, int R, int NW, int NTW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void rowgemm_xs_kernel(RowGemmArgs a) {
    constexpr int KK = KP / 16;
    constexpr int CH = KP <= 192 ? 1 : 2;
    constexpr int PD = 2;                       // weight ring depth in k-steps (NTW fragments each)
    static_assert(R <= NW, "one wave per row tile in the LayerNorm phase");
    extern __shared__ f32x4 xl[];               // [R][KK][64] normalised rows, operand layout
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = blockIdx.x * (16 * R);
    const int nt0 = wave * NTW;

    // this wave's first weight fragments are on their way before anything else
    const f32x4* wsrc[NTW];
    f32x4 ring[PD][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        wsrc[j] = a.wf + (size_t)min(nt0 + j, a.NT - 1) * KK * 64 + lane;       // ragged tail: a duplicate tile, never stored
#pragma unroll
        for (int i = 0; i < PD; ++i) ring[i][j] = wsrc[j][i * 64];
    }
    if (wave < R) {
        f32x4 xf[KK];
        rowgemm_gather<KP, SEGS>(a, row0 + wave * 16, l15, lg, xf);
        rowgemm_normalise<KP, SEGS>(a, lg, xf);
        f32x4* dst = xl + (size_t)wave * KK * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) dst[kk * 64] = xf[kk];
    }
    __syncthreads();

    f32x4 acc[NTW][R][CH];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[j][t][c] = zero4();
    const f32x4* xb = xl + lane;
    f32x4 xn[R];                                // row fragments one k-step ahead
#pragma unroll
    for (int t = 0; t < R; ++t) xn[t] = xb[(t * KK) * 64];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        f32x4 x[R];
#pragma unroll
        for (int t = 0; t < R; ++t) { x[t] = xn[t]; if (kk + 1 < KK) xn[t] = xb[(t * KK + kk + 1) * 64]; }
        f32x4 w[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            w[j] = ring[kk % PD][j];
            if (kk + PD < KK) ring[kk % PD][j] = wsrc[j][(kk + PD) * 64];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const int c = (CH == 2) ? (r & 1) : 0;
                    acc[j][t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][r], x[t][r], acc[j][t][c], 0, 0, 0);
                }
        // pin the software pipeline: unpinned, hipcc sinks every weight load to just before its first use and waits out the L2 round trip
        if (kk + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
        if (kk + PD < KK) __builtin_amdgcn_sched_group_barrier(0x020, NTW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * R * NTW, 0);
    }
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int row = row0 + t * 16 + l15;
        if (row >= a.M) continue;
        size_t ob[2];
        rowgemm_obase(a, row, ob);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (nt0 + j >= a.NT) continue;
            rowgemm_store(a, ob, 16 * (nt0 + j) + 4 * lg, CH == 2 ? acc[j][t][0] + acc[j][t][1] : acc[j][t][0]);
        }
    }
}

#endif  // ESCX_EXPERIMENTAL

}  // namespace escx
