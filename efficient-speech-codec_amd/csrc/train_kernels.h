// Device code of the TRAINING step (reference: scripts/trainer_no_adv.py:95-118, esc/modules/vq/codebook.py:57-75,
// esc/modules/vq/quantization.py:31-72, esc/models/csrvq.py:23-48, esc/modules/loss/generator_loss.py:12-74).
//
// The forward of a training step reuses the plain GEMM pipeline of the inference path (LN -> QKV GEMM -> window attention -> proj GEMM -> LN ->
// fc1 -> fc2) with every activation a backward pass needs kept in HBM (288 GB: nothing is recomputed except LayerNorm statistics and the
// softmax rows).  This file holds what only training needs:
//   * gemm_dw_kernel      dW[n][k] = sum_m dY[m][n] * X[m][k]  (+ db[n]) on the fp32 MFMA, M split over workgroups, partial sums reduced in a FIXED
//                         order by reduce_partials_kernel (no atomics anywhere: gradients are run-to-run deterministic)
//   * ln_bwd_kernel       LayerNorm backward (statistics recomputed), window-gather / merge-gather aware, dgamma/dbeta as fixed-order partials
//   * attn_bwd_kernel     4x4-window attention core backward (softmax recomputed), relative-position-bias gradient
//   * loaders / epilogues for the dX GEMMs (gemm_engine.h's kernel with transposed weights)
//   * product-VQ training forward/backward (straight-through estimator, commitment / codebook losses, embedding gradient by code)
//   * loss kernels (power-law complex STFT loss, multi-scale mel loss pieces), inverse-STFT backward, AdamW / global-norm clipping
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"

namespace escx {

// ------------------------------------------------------------------------------------------------
// extra A-side loaders (row m, 4 consecutive columns k0+kin..)
// ------------------------------------------------------------------------------------------------
struct SlotGatherA {            // rows in window-slot order gathered from a token-major map: A[(b,slot)][k] = x[(b, map[slot])][k], pad slots = 0
    const float* x; const int* map; int slots, tokens, ld, M; FastDiv dSlots;
    typedef const float* Ctx;
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        if (m >= M) return nullptr;
        const int b = dSlots.div(m), s = m - b * slots;
        const int tok = map[s];
        return tok < 0 ? nullptr : x + ((size_t)b * tokens + tok) * ld;
    }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const { return c ? ld4(c + k0 + kin) : zero4(); }
};

struct SplitGatherA {           // inverse of the PatchSplit pixel shuffle: A[(b,h,w)][s*C2p + c] = g[(b, 2h+s, w)][c]
    const float* g; int H, W, C2p, M; FastDiv dHW, dW, dC2p;
    typedef int Ctx;            // element offset of token (b, 2h, w), or -1
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        if (m >= M) return -1;
        const int b = dHW.div(m), r = m - b * H * W; const int h = dW.div(r), w = r - h * W;
        return ((b * 2 * H + 2 * h) * W + w) * C2p;
    }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const {
        if (c < 0) return zero4();
        const int k = k0 + kin; const int s = dC2p.div(k), cc = k - s * C2p;
        if (s > 1) return zero4();
        return ld4(g + (size_t)c + (size_t)s * W * C2p + cc);
    }
};

struct ShuffleA {               // inverse of the de-embedding pixel shuffle: A[(b,h,w)][q*Cp + c] = g[(b, t=pt*w+s2, f=pf*h+s1)][c], q = s1*pt+s2
    const float* g; int H, W, Cp, pf, pt, M; FastDiv dHW, dW, dCp;      // g is the TIME-major fine map [b][pt*W][pf*H][Cp]
    struct Ctx { int b, h, w; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.b = -1; c.h = 0; c.w = 0;
        if (m < M) { c.b = dHW.div(m); const int r = m - c.b * H * W; c.h = dW.div(r); c.w = r - c.h * W; }
        return c;
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (c.b < 0) return zero4();
        const int k = k0 + kin; const int q = dCp.div(k), cc = k - q * Cp;
        if (q >= pf * pt) return zero4();
        const int s1 = q / pt, s2 = q - s1 * pt;
        return ld4(g + (((size_t)c.b * (pt * W) + pt * c.w + s2) * (pf * H) + pf * c.h + s1) * Cp + cc);
    }
};

template <class Ld>
struct ColOffset {              // columns [off, off + width) of another loader's rows (the diagonal blocks of a block-structured contraction)
    Ld ld; int off;
    typedef typename Ld::Ctx Ctx;
    __device__ __forceinline__ Ctx make_ctx(int m) const { return ld.make_ctx(m); }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const { return ld.load4(c, k0 + off, kin); }
};

struct ConvShuffleA {           // conv5x5 dX: A[(b,h,w)][k = (tap, q, c)] = dY1[(b, h-dh, w-dw)][q*Cp+c] with dY1 = ShuffleA(g); tap = (kh,kw), dh = kh-2
    const float* g; int H, W, Cp, pf, pt, M; FastDiv dHW, dW, dCp, dQ;
    struct Ctx { int b, h, w; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.b = -1; c.h = 0; c.w = 0;
        if (m < M) { c.b = dHW.div(m); const int r = m - c.b * H * W; c.h = dW.div(r); c.w = r - c.h * W; }
        return c;
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (c.b < 0) return zero4();
        const int k = k0 + kin; const int tq = dCp.div(k), cc = k - tq * Cp;
        const int tap = dQ.div(tq), q = tq - tap * (pf * pt);
        if (tap >= 25) return zero4();
        const int kh = tap / 5, kw = tap - kh * 5;
        const int hs = c.h - (kh - 2), ws = c.w - (kw - 2);          // forward: out(h,w) reads in(h+kh-2, w+kw-2)  =>  in(h,w) feeds out(h-(kh-2), w-(kw-2))
        if (hs < 0 || hs >= H || ws < 0 || ws >= W) return zero4();
        const int s1 = q / pt, s2 = q - s1 * pt;
        return ld4(g + (((size_t)c.b * (pt * W) + pt * ws + s2) * (pf * H) + pf * hs + s1) * Cp + cc);
    }
};

struct SpecRowsA {              // rows m = (b, t, f) of a frame-major spectrum gradient: A[m][oc] = g[(b,t)][oc*Fp + f], oc < in_dim
    const float* g; int F, Fp, in_dim, M; FastDiv dF;
    typedef const float* Ctx;
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        if (m >= M) return nullptr;
        const int bt = dF.div(m), f = m - bt * F;
        return g + (size_t)bt * (in_dim * Fp) + f;
    }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const {
        f32x4 v = zero4();
        if (!c) return v;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (k0 + kin + r < in_dim) v[r] = c[(size_t)(k0 + kin + r) * Fp];
        return v;
    }
};

// ------------------------------------------------------------------------------------------------
// extra epilogues
// ------------------------------------------------------------------------------------------------
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x), branch-free: erf_bf (1.1 ulp, gemm_engine.h) and v_exp_f32 instead of libm's erff / expf, whose
// data-dependent branches made this epilogue cost more VALU time than the GEMM it rides on costs MFMA time at C <= 96
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erf_bf(x * 0.70710678118654752440f));
    return fmaf(x * 0.39894228040143267794f, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f), cdf);
}

// gelu(x) and gelu'(x) from ONE exponential chain (the fused MLP backward evaluates both for every hidden unit): with E = Phi(-|x|) = 2^Q8(|x|) / 2
// of gelu_bf (gemm_engine.h), gelu = max(x, 0) - |x| E, Phi(x) = x >= 0 ? 1 - E : E and gelu' = Phi + x phi(x).  |gelu' error| <= 4.4e-7 over the
// whole line (fp32 evaluation of the exact formula: 1.1e-7), 18 VALU ops for the pair instead of 11 + 35.
__device__ __forceinline__ void gelu_pair(float x, float& act, float& grad) {
    const float a = fabsf(x);
    float r = -1.6904631365832756e-06f;
    r = fmaf(r, a, 2.5084045773837715e-05f);
    r = fmaf(r, a, -0.0001144662601291202f);
    r = fmaf(r, a, -0.0003233331080991775f);
    r = fmaf(r, a, 0.007333371322602034f);
    r = fmaf(r, a, -0.052714187651872635f);
    r = fmaf(r, a, -0.4591154456138611f);
    r = fmaf(r, a, -1.151123285293579f);
    r = fmaf(r, a, 1.126017423302983e-06f - 1.0f);
    const float E = __builtin_amdgcn_exp2f(r);
    act = fmaf(-a, E, fmaxf(x, 0.0f));
    const float cdf = x >= 0.0f ? 1.0f - E : E;
    grad = fmaf(x * 0.39894228040143267794f, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f), cdf);
}

struct EpiGeluDual {            // fc1 in training: pre-activation AND activation are kept (attention.py:267-272)
    float* pre; float* act; int ldo; const float* bias;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        v += ld4(bias + n);
        st4(pre + (size_t)m * ldo + n, v);
        v[0] = gelu_bf(v[0]); v[1] = gelu_bf(v[1]); v[2] = gelu_bf(v[2]); v[3] = gelu_bf(v[3]);       // the fused inference kernels' exact-erf GELU (11 VALU ops)
        st4(act + (size_t)m * ldo + n, v);
    }
};

struct EpiGeluBwd {             // dh_pre = (dy . W2) * gelu'(h_pre)
    float* out; int ldo; const float* pre;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const f32x4 p = ld4(pre + (size_t)m * ldo + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float ac, gr; gelu_pair(p[e], ac, gr); v[e] *= gr; }      // the exponent-chain form of the fused backward (the unused value is dead code): ~16 ops instead of ~32
        st4(out + (size_t)m * ldo + n, v);
    }
};

struct EpiAccum {               // out[m][n] += v  (gradient accumulation where a tensor has two consumers)
    float* out; int ldo;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        float* p = out + (size_t)m * ldo + n;
        st4(p, ld4(p) + v);
    }
};

struct EpiPvqGrad {             // gradient of the framed residual back on the token maps: d_enc += v, d_dec -= v  (csrvq.py:15-17)
    float* denc; float* ddec; int Hq, W, Cp, Tq, ov; FastDiv dTq, dCp, dHq;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const int b = dTq.div(m), t = m - b * Tq;
        const int oh = dCp.div(n), c = n - oh * Cp;
        const int o = dHq.div(oh), h = oh - o * Hq;
        const size_t idx = ((size_t)(b * Hq + h) * W + ov * t + o) * Cp + c;
        st4(denc + idx, ld4(denc + idx) + v);
        if (ddec) st4(ddec + idx, ld4(ddec + idx) - v);
    }
};

// LayerNorm backward as the ROW EPILOGUE of the GEMM that produces its upstream gradient (dx_fc1: d LN2(x1) = d h_pre . W1), for maps whose
// channel count fits one workgroup tile (Cp <= 96): out = add + LNbwd(acc; x), written in token order and - through slot_of - in window-slot
// order; the standalone pass re-read the GEMM's output, x and add and was HBM-bound.  dgamma / dbeta leave as one partial row per wave
// (part[(workgroup*4 + wave)][2][Cp]), added in a fixed order by reduce_partials.
struct EpiLnBwdRows {
    static constexpr bool ROWWISE = true;
    const float* x; const float* gamma; const float* add; float* dx; float* dx_slots; const int* slot_of; float* part;
    int C, Cp, rows_per_clip, slots_per_clip; float eps;
    const int* row_map;         // optional: the GEMM's rows are window SLOTS (LN1: the upstream gradient comes out of the QKV GEMM in slot order);
                                // row_map[slot] = token of the clip or -1 for a pad slot, x / add / dx are indexed by token
    __device__ __forceinline__ void store(int, int, f32x4, int) const {}
    // Wide maps (Cp = 144 / 192 / 384, round 4): the narrow form below keeps x, gamma and both column-sum accumulators of a whole row tile in
    // registers (4 x TN float4 next to the accumulators) - at TN = 12..24 that is beyond the register file.  Here phase 1 takes the row statistics and
    // the two projection sums per row tile (x held for ONE row tile at a time), phase 2 walks the column tiles, re-reads x (an L2 hit) and finishes
    // each column tile's rows and column sums before the next: the same per-element arithmetic in the same order, accumulators + ~60 registers.
    template <int TN, int TM>
    __device__ __forceinline__ void finish_wide(f32x4 (&acc)[TN][TM], int row0, int lane, int M) const {
        const int l15 = lane & 15, lg = lane >> 4;
        const float invC = 1.0f / (float)C;
        int mrow[TM], dsrow[TM]; bool lv[TM]; float mean[TM], rstd[TM], c1[TM], c2[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            int m = row0 + 16 * b + l15;
            bool live = m < M;
            if (row_map && live) {
                const int bi = m / slots_per_clip; const int tok = row_map[m - bi * slots_per_clip];
                live = tok >= 0; m = bi * rows_per_clip + tok;
            }
            mrow[b] = live ? m : 0; lv[b] = live; dsrow[b] = -1;
            if (live && dx_slots) { const int bi = m / rows_per_clip, rr = m - bi * rows_per_clip; dsrow[b] = bi * slots_per_clip + slot_of[rr]; }
            f32x4 xv[TN];
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = 16 * a + 4 * lg;
                xv[a] = (live && n < Cp) ? ld4(x + (size_t)m * Cp + n) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < C) s += xv[a][e];
            }
            mean[b] = sum_groups(s) * invC;
            float var = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (16 * a + 4 * lg + e < C) { const float d = xv[a][e] - mean[b]; var += d * d; }
            rstd[b] = 1.0f / sqrtf(sum_groups(var) * invC + eps);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = 16 * a + 4 * lg;
                const f32x4 gm = n < Cp ? ld4(gamma + n) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < C) { const float xh = (xv[a][e] - mean[b]) * rstd[b], t = acc[a][b][e] * gm[e]; s1 += t; s2 += t * xh; }
            }
            c1[b] = sum_groups(s1) * invC; c2[b] = sum_groups(s2) * invC;
        }
        float* pr = part + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 * Cp;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = 16 * a + 4 * lg;
            const f32x4 gm = n < Cp ? ld4(gamma + n) : zero4();
            f32x4 ag = zero4(), ab = zero4();
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                if (!lv[b] || n >= Cp) continue;
                const f32x4 xr = ld4(x + (size_t)mrow[b] * Cp + n);
                f32x4 o = add ? ld4(add + (size_t)mrow[b] * Cp + n) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xr[e] - mean[b]) * rstd[b], g = acc[a][b][e];
                    if (n + e < C) { ag[e] += g * xh; ab[e] += g; o[e] = o[e] + rstd[b] * (g * gm[e] - c1[b] - xh * c2[b]); }
                    else o[e] = 0.f;
                }
                st4(dx + (size_t)mrow[b] * Cp + n, o);
                if (dsrow[b] >= 0) st4(dx_slots + (size_t)dsrow[b] * Cp + n, o);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = ag[e], w = ab[e];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o, 16); w += __shfl_xor(w, o, 16); }
                ag[e] = u; ab[e] = w;
            }
            if (l15 == 0 && n < Cp) { st4(pr + n, ag); st4(pr + Cp + n, ab); }
        }
    }
    template <int TN, int TM>
    __device__ __forceinline__ void finish(f32x4 (&acc)[TN][TM], int row0, int lane, int M) const {
        if constexpr (TN > 6) { finish_wide<TN, TM>(acc, row0, lane, M); return; }
        const int l15 = lane & 15, lg = lane >> 4;
        const float invC = 1.0f / (float)C;
        f32x4 gm[TN], ag[TN], ab[TN];
#pragma unroll
        for (int a = 0; a < TN; ++a) { const int n = 16 * a + 4 * lg; gm[a] = n < Cp ? ld4(gamma + n) : zero4(); ag[a] = zero4(); ab[a] = zero4(); }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            int m = row0 + 16 * b + l15;
            bool live = m < M;
            if (row_map && live) {
                const int bi = m / slots_per_clip; const int tok = row_map[m - bi * slots_per_clip];
                live = tok >= 0; m = bi * rows_per_clip + tok;
            }
            f32x4 xv[TN];
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = 16 * a + 4 * lg;
                xv[a] = (live && n < Cp) ? ld4(x + (size_t)m * Cp + n) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < C) s += xv[a][e];
            }
            const float mean = sum_groups(s) * invC;
            float var = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (16 * a + 4 * lg + e < C) { const float d = xv[a][e] - mean; var += d * d; }
            const float rstd = 1.0f / sqrtf(sum_groups(var) * invC + eps);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (16 * a + 4 * lg + e < C) {
                        const float xh = (xv[a][e] - mean) * rstd, g = acc[a][b][e], t = g * gm[a][e];
                        s1 += t; s2 += t * xh;
                        if (live) { ag[a][e] += g * xh; ab[a][e] += g; }
                    }
            const float c1 = sum_groups(s1) * invC, c2 = sum_groups(s2) * invC;
            if (live) {
                const int bi = m / rows_per_clip, rr = m - bi * rows_per_clip;
                float* ds = dx_slots ? dx_slots + ((size_t)bi * slots_per_clip + slot_of[rr]) * Cp : nullptr;
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int n = 16 * a + 4 * lg;
                    if (n >= Cp) continue;
                    f32x4 o = add ? ld4(add + (size_t)m * Cp + n) : zero4();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (xv[a][e] - mean) * rstd;
                        o[e] = (n + e < C) ? o[e] + rstd * (acc[a][b][e] * gm[a][e] - c1 - xh * c2) : 0.f;
                    }
                    st4(dx + (size_t)m * Cp + n, o);
                    if (ds) st4(ds + n, o);
                }
            }
        }
        // column sums over this wave's rows: the 16 lanes of a lane group hold 16 different rows of the same columns
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = ag[a][e], w = ab[a][e];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o, 16); w += __shfl_xor(w, o, 16); }
                ag[a][e] = u; ab[a][e] = w;
            }
        if (l15 == 0) {
            float* pr = part + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 * Cp;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = 16 * a + 4 * lg;
                if (n < Cp) { st4(pr + n, ag[a]); st4(pr + Cp + n, ab[a]); }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// dW[n][k] = sum_m A[m][n] * B[m][k]   (A = upstream gradient rows, B = saved input rows), optional db[n] = sum_m A[m][n].
// One workgroup = one 48 x 48 output tile and one slice of M; its 4 waves take alternate 32-row chunks, stage them in their own LDS
// region and run 9 MFMAs per 4 rows; the waves are then added in the order 0,1,2,3 and the slice's partial tile is written to
// part[slice][Np][Kp].  reduce_partials_kernel adds the slices in increasing order.
// ------------------------------------------------------------------------------------------------
constexpr int DW_T = 3;                 // 16-wide tiles per side of the workgroup tile
constexpr int DW_LD = 48;               // LDS row stride in floats: 48 % 32 == 16 -> the 4 rows of an MFMA operand read hit different bank groups
constexpr int DW_MC = 32;               // rows per staged chunk

template <class LdA, class LdB, bool BIAS>
__global__ __launch_bounds__(256) void gemm_dw_kernel(LdA la, LdB lb, int M, int Np, int Kp, int nblk_k, int m_per_slice,
                                                      float* __restrict__ part, float* __restrict__ bpart) {
    __shared__ float lds[4 * 2 * DW_MC * DW_LD];       // per wave: A chunk | B chunk  (4 * 2 * 32 * 48 * 4 B = 48 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int bn = blockIdx.x / nblk_k, bk = blockIdx.x - bn * nblk_k;
    const int n0 = bn * 16 * DW_T, k0 = bk * 16 * DW_T;
    const int mbeg = blockIdx.y * m_per_slice, mend = min(M, mbeg + m_per_slice);
    float* As = lds + wave * (2 * DW_MC * DW_LD);
    float* Bs = As + DW_MC * DW_LD;
    // Bias gradient = column sums of the dY rows.  Rounds 1-3 had it ride on the MFMA (a column of ones as a fourth B tile, under `if (do_bias)`): the
    // conditional matrix instructions inside the unrolled loop made the register allocator shuttle every accumulator between AGPRs and VGPRs - 2
    // v_accvgpr copies per MFMA, 5x the VALU instructions of the bias-free instantiation (ISA counts in DESIGN 7a).  Now every lane adds the float4s it
    // stages (its column group is the same in every chunk), and the lanes / waves are added in a fixed order through LDS at the end.
    const bool do_bias = BIAS && bk == 0;

    f32x4 acc[DW_T][DW_T];
#pragma unroll
    for (int a = 0; a < DW_T; ++a)
#pragma unroll
        for (int b = 0; b < DW_T; ++b) acc[a][b] = zero4();
    f32x4 sa[BIAS ? 6 : 1];
#pragma unroll
    for (int j = 0; j < (BIAS ? 6 : 1); ++j) sa[j] = zero4();

    // each lane moves 6 float4 per operand per chunk: element e -> (row e / 12, float4 column e % 12).  The NEXT chunk's global loads are
    // issued before the current chunk's MFMAs (register prefetch): one memory round trip per chunk is hidden behind 72 MFMAs.
    constexpr int V = DW_T * 4;          // float4 per staged row
    f32x4 ra[6], rb[6];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int e = lane + 64 * j; const int row = e / V, c4 = e - row * V;
            const int m = m0 + row;
            const int col = 4 * c4;                                  // column inside the 48-wide tile
            const bool live = m < mend;
            typename LdA::Ctx ca = la.make_ctx(live ? m : M);
            typename LdB::Ctx cb = lb.make_ctx(live ? m : M);
            ra[j] = (n0 + col < Np) ? la.load4(ca, n0 + (col & ~15), col & 15) : zero4();
            rb[j] = (k0 + col < Kp) ? lb.load4(cb, k0 + (col & ~15), col & 15) : zero4();
        }
    };
    const int mfirst = mbeg + wave * DW_MC;
    if (mfirst < mend) fetch(mfirst);
    for (int m0 = mfirst; m0 < mend; m0 += 4 * DW_MC) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int e = lane + 64 * j; const int row = e / V, c4 = e - row * V;
            st4(As + row * DW_LD + 4 * c4, ra[j]);
            st4(Bs + row * DW_LD + 4 * c4, rb[j]);
            if constexpr (BIAS) sa[j] += ra[j];          // rows beyond the slice were fetched as zeros
        }
        if (m0 + 4 * DW_MC < mend) fetch(m0 + 4 * DW_MC);
        // a wave only reads what it wrote itself: no workgroup barrier, the LDS queue is in order per wave
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ms = 0; ms < DW_MC / 4; ++ms) {
            float af[DW_T], bf[DW_T];
#pragma unroll
            for (int a = 0; a < DW_T; ++a) af[a] = As[(4 * ms + lg) * DW_LD + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < DW_T; ++b) bf[b] = Bs[(4 * ms + lg) * DW_LD + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < DW_T; ++a)
#pragma unroll
                for (int b = 0; b < DW_T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // column sums: every wave folds its 32 staged rows in its OWN staging area (rows in order), then the four waves are added in order
    __shared__ float redb[4][16 * DW_T];
    if constexpr (BIAS) {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 6; ++j) { const int e = lane + 64 * j; const int row = e / V, c4 = e - row * V; st4(As + row * DW_LD + 4 * c4, sa[j]); }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (lane < 16 * DW_T) {
                float t = 0.f;
                for (int r = 0; r < DW_MC; ++r) t += As[r * DW_LD + lane];
                redb[wave][lane] = t;
            }
        }
    }
    // add the four waves in the order 0,1,2,3 (fixed), wave 0 writes the slice's partial tile
    __syncthreads();
    if constexpr (BIAS) {
        if (do_bias && tid < 16 * DW_T && n0 + tid < Np)
            bpart[(size_t)blockIdx.y * Np + n0 + tid] = ((redb[0][tid] + redb[1][tid]) + redb[2][tid]) + redb[3][tid];
    }
    float* red = lds;                                                // [3 waves][9 tiles][64 lanes][4] = 27 KB of the 48 KB staging area
    if (wave > 0) {
        float* r = red + (size_t)(wave - 1) * 9 * 256;
#pragma unroll
        for (int a = 0; a < DW_T; ++a) {
#pragma unroll
            for (int b = 0; b < DW_T; ++b) st4(r + ((a * DW_T + b) * 64 + lane) * 4, acc[a][b]);
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* r = red + (size_t)w * 9 * 256;
#pragma unroll
            for (int a = 0; a < DW_T; ++a) {
#pragma unroll
                for (int b = 0; b < DW_T; ++b) acc[a][b] += ld4(r + ((a * DW_T + b) * 64 + lane) * 4);
            }
        }
        float* po = part + (size_t)blockIdx.y * Np * Kp;
#pragma unroll
        for (int a = 0; a < DW_T; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * a + 4 * lg + r;
                if (n >= Np) continue;
#pragma unroll
                for (int b = 0; b < DW_T; ++b) {
                    const int k = k0 + 16 * b + l15;
                    if (k < Kp) po[(size_t)n * Kp + k] = acc[a][b][r];
                }
            }
    }
}

// The contraction through arbitrary row loaders with a WIDE workgroup tile: WN x WK waves, each owning TA x TB accumulator tiles, share every
// staged 32-row chunk through a double-buffered LDS image (one barrier per chunk, next chunk's loads in flight during the MFMAs).
//   <4,4,2,2>  128 x 128 tile for large weight matrices (N, K >= 256: the 512 / 1024-channel convolutions of the discriminator): 256 floats
//              staged per row for 16384 multiply-adds (the 48 x 48 kernel: 96 floats for 2304 - 2.7x the operand traffic and gather arithmetic)
//   <2,3,1,4>   32 x 192 tile for N = 32 (the 32-channel band stacks of MRD): no padded third of the N side, dY staged once for four K tiles
// Every thread owns ONE row of the chunk (8 threads per row): one make_ctx per operand per chunk.
template <class LdA, class LdB, int TA, int TB, int WN, int WK, bool BIAS>
__global__ __launch_bounds__(256) void gemm_dw3_kernel(LdA la, LdB lb, int M, int Np, int Kp, int nblk_k, int m_per_slice,
                                                       float* __restrict__ part, float* __restrict__ bpart) {
    static_assert(WN * WK == 4, "four waves");
    constexpr int WA = 16 * TA * WN, WB = 16 * TB * WK;          // workgroup tile (sides are multiples of 16: 144 and 80 for the C = 144 / 72 maps)
    constexpr int LDA = WA % 32 == 0 ? WA + 16 : WA, LDB = WB % 32 == 0 ? WB + 16 : WB;      // % 32 == 16: the 4 rows of an MFMA operand read hit different bank groups
    constexpr int NA = (WA + 31) / 32, NB = (WB + 31) / 32;      // float4 per thread per row (8 threads per row; the last one partly masked when the side is not a multiple of 32)
    __shared__ float lds[2 * DW_MC * (LDA + LDB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wn = wave / WK, wk = wave - wn * WK;
    const int bn = blockIdx.x / nblk_k, bk = blockIdx.x - bn * nblk_k;
    const int n0 = bn * WA, k0 = bk * WB;
    const int mbeg = blockIdx.y * m_per_slice, mend = min(M, mbeg + m_per_slice);
    const bool do_bias = BIAS && bk == 0;                       // bias gradient = column sums of the staged dY rows, on the VALU (see gemm_dw_kernel)
    const int frow = tid >> 3, fc = tid & 7;
    f32x4 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = zero4();
    f32x4 ra[NA], rb[NB];
    f32x4 sa[BIAS ? NA : 1];
#pragma unroll
    for (int j = 0; j < (BIAS ? NA : 1); ++j) sa[j] = zero4();
    auto fetch = [&](int m0) {
        const int m = m0 + frow;
        typename LdA::Ctx ca = la.make_ctx(m < mend ? m : M);
        typename LdB::Ctx cb = lb.make_ctx(m < mend ? m : M);
#pragma unroll
        for (int j = 0; j < NA; ++j) { const int col = 4 * (fc + 8 * j); ra[j] = (col < WA && n0 + col < Np) ? la.load4(ca, n0 + (col & ~15), col & 15) : zero4(); }
#pragma unroll
        for (int j = 0; j < NB; ++j) { const int col = 4 * (fc + 8 * j); rb[j] = (col < WB && k0 + col < Kp) ? lb.load4(cb, k0 + (col & ~15), col & 15) : zero4(); }
    };
    if (mbeg < mend) fetch(mbeg);
    int buf = 0;
    for (int m0 = mbeg; m0 < mend; m0 += DW_MC) {
        float* As = lds + buf * (DW_MC * (LDA + LDB));
        float* Bs = As + DW_MC * LDA;
#pragma unroll
        for (int j = 0; j < NA; ++j) if (WA % 32 == 0 || 4 * (fc + 8 * j) < WA) { st4(As + frow * LDA + 4 * (fc + 8 * j), ra[j]); if constexpr (BIAS) sa[j] += ra[j]; }
#pragma unroll
        for (int j = 0; j < NB; ++j) if (WB % 32 == 0 || 4 * (fc + 8 * j) < WB) st4(Bs + frow * LDB + 4 * (fc + 8 * j), rb[j]);
        __syncthreads();                                         // one barrier per chunk: the other buffer was last read before the previous barrier
        if (m0 + DW_MC < mend) fetch(m0 + DW_MC);
        const float* Aw = As + wn * 16 * TA; const float* Bw = Bs + wk * 16 * TB;
#pragma unroll
        for (int ms = 0; ms < DW_MC / 4; ++ms) {
            float af[TA], bf[TB];
#pragma unroll
            for (int a = 0; a < TA; ++a) af[a] = Aw[(4 * ms + lg) * LDA + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < TB; ++b) bf[b] = Bw[(4 * ms + lg) * LDB + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        buf ^= 1;
    }
    if constexpr (BIAS) {
        if (do_bias) {                                           // uniform per workgroup: rows 0..31 of the thread grid in order
            __syncthreads();                                     // every wave is done with the staging buffers
#pragma unroll
            for (int j = 0; j < NA; ++j) if (WA % 32 == 0 || 4 * (fc + 8 * j) < WA) st4(lds + frow * LDA + 4 * (fc + 8 * j), sa[j]);
            __syncthreads();
            for (int n = tid; n < WA; n += 256) {
                float t = 0.f;
                for (int r = 0; r < DW_MC; ++r) t += lds[r * LDA + n];
                if (n0 + n < Np) bpart[(size_t)blockIdx.y * Np + n0 + n] = t;
            }
        }
    }
    float* po = part + (size_t)blockIdx.y * Np * Kp;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 16 * TA + 16 * a + 4 * lg + r;
            if (n >= Np) continue;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int k = k0 + wk * 16 * TB + 16 * b + l15;
                if (k < Kp) po[(size_t)n * Kp + k] = acc[a][b][r];
            }
        }
}

// out[i] = sum_s part[s][i], s increasing (fixed order); optional accumulate into out
static __global__ void reduce_partials_kernel(const float* __restrict__ part, int slices, long long n, float* __restrict__ out, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < slices; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}
// The same for MANY slices of a SMALL output (a 48 x 48 weight gradient summed over ~2000 slices): a thread per output would walk its
// slices as one long chain of dependent L2 loads (~1 ms); here a 64-lane wave owns 16 consecutive outputs, lane (q, j) adds slices
// q, q+4, q+8, ... of output j (coalesced 64 B reads), then the 4 partial sums are joined in the fixed order ((0+1)+(2+3)).
static __global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ part, int slices, long long n, float* __restrict__ out,
                                                                   int accumulate) {
    const long long wv = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const long long i = wv * 16 + j;
    float s = 0.f;
    if (i < n) for (int k = q; k < slices; k += 4) s += part[(size_t)k * n + i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (i < n && q == 0) out[i] = (accumulate ? out[i] : 0.f) + s;
}

// For MANY slices (>= 128): 16 outputs x 16 slice lanes per workgroup - lane (q, j) adds slices q, q+16, ... of output j, the 16 partial sums
// are then added in the order 0..15 by lane q = 0.  Four times the loads in flight of the wave-per-16-outputs form above, which still walked
// ~500 dependent L2 loads per lane for the 48-channel weight gradients (2000 slices) and the 512-slice bias gradients of the discriminator.
static __global__ __launch_bounds__(256) void reduce_partials_tree_kernel(const float* __restrict__ part, int slices, long long n, float* __restrict__ out,
                                                                   int accumulate) {
    __shared__ float red[16][17];
    const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + j;
    float s = 0.f;
    if (i < n) {
        int k = q;
        for (; k + 48 < slices; k += 64) {                     // four independent loads per round trip
            const float a = part[(size_t)k * n + i], b = part[(size_t)(k + 16) * n + i], c = part[(size_t)(k + 32) * n + i], d = part[(size_t)(k + 48) * n + i];
            s += a; s += b; s += c; s += d;
        }
        for (; k < slices; k += 16) s += part[(size_t)k * n + i];
    }
    red[q][j] = s;
    __syncthreads();
    if (q == 0 && i < n) {
        float t = red[0][j];
#pragma unroll
        for (int r = 1; r < 16; ++r) t += red[r][j];
        out[i] = (accumulate ? out[i] : 0.f) + t;
    }
}
// fixed-order sum of `slices` partial tensors of n floats.  The variant depends on (slices, n) only - never on data - so results stay
// run-to-run deterministic.
// slices [g*per, (g+1)*per) of a [slices][n] tensor -> tmp[g][n]   (first stage of the two-stage reduction below; grid = (n/16, groups))
static __global__ __launch_bounds__(256) void reduce_partials_stage_kernel(const float* __restrict__ part, int slices, int per, long long n, float* __restrict__ tmp) {
    __shared__ float red[16][17];
    const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + j;
    const int k0 = blockIdx.y * per, k1 = min(slices, k0 + per);
    float s = 0.f;
    if (i < n) for (int k = k0 + q; k < k1; k += 16) s += part[(size_t)k * n + i];
    red[q][j] = s;
    __syncthreads();
    if (q == 0 && i < n) {
        float t = red[0][j];
#pragma unroll
        for (int r = 1; r < 16; ++r) t += red[r][j];
        tmp[(size_t)blockIdx.y * n + i] = t;
    }
}
static inline void launch_reduce_partials(const float* part, int slices, long long n, float* out, int accumulate, hipStream_t st, float* tmp = nullptr) {
    auto nb = [](long long v) { return (unsigned)((v + 255) / 256); };
    if (tmp && slices >= 2048 && n <= 4096) {                  // very many slices of a short row: 64 groups first (tmp: 64 * n floats), then the 64
        const int groups = 64, per = (slices + groups - 1) / groups;
        hipLaunchKernelGGL(reduce_partials_stage_kernel, dim3((unsigned)((n + 15) / 16), groups), dim3(256), 0, st, part, slices, per, n, tmp);
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3(nb((n + 15) / 16 * 64)), dim3(256), 0, st, tmp, (slices + per - 1) / per, n, out, accumulate);
        return;
    }
    if (slices >= 128 && n <= ((long long)1 << 18))
        hipLaunchKernelGGL(reduce_partials_tree_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, slices, n, out, accumulate);
    else if (slices >= 32 && n * 4 <= ((long long)1 << 22))
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3(nb((n + 15) / 16 * 64)), dim3(256), 0, st, part, slices, n, out, accumulate);
    else
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(nb(n)), dim3(256), 0, st, part, slices, n, out, accumulate);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  Modes as in ln_rows_kernel; statistics are recomputed from the saved input.
//   MODE 0: rows = tokens, dy row = row.                         dx[row] = (add ? add[row] : 0) + dLN
//   MODE 1: rows = tokens, dy row = slot inv[token] of the clip (the forward gathered tokens into window slots; pad slots carry no gradient)
//   MODE 2: rows = merged rows (2 segments gathered through map); dx goes to the two source tokens (each is used exactly once)
// dgamma / dbeta: every thread accumulates its channels over the rows it walks, the 16 row groups of a workgroup are added in a fixed
// order in LDS, and the workgroup writes one partial row; reduce_partials_kernel finishes.  part: [gridDim.x][2][SEGS*Cp]
// ------------------------------------------------------------------------------------------------
template <int SEGS, int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                     const int* __restrict__ map, const float* __restrict__ add, float* __restrict__ dx,
                                                     float* __restrict__ part, int rows_per_clip, int src_rows_per_clip, int dy_rows_per_clip,
                                                     int total_rows, int C, int Cp, float eps, float* __restrict__ dx_slots, const int* __restrict__ slot_of,
                                                     int slots_per_clip) {
    constexpr int MAXV = 6;                       // float4 per thread per segment: Cp <= 384
    extern __shared__ float dyn[];                // [16 groups][2][SEGS*Cp]
    const int sub = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int ngrp = (gridDim.x * 256) >> 4;
    const int V = Cp / 4;
    f32x4 ag[SEGS][MAXV], ab[SEGS][MAXV];
#pragma unroll
    for (int s = 0; s < SEGS; ++s)
#pragma unroll
        for (int q = 0; q < MAXV; ++q) { ag[s][q] = zero4(); ab[s][q] = zero4(); }

    for (int row = grp; row < total_rows; row += ngrp) {
        const int b = row / rows_per_clip, rr = row - b * rows_per_clip;
        const float* sp[SEGS]; float* dp[SEGS];
        const float* gp;
        if (MODE == 2) {
#pragma unroll
            for (int s = 0; s < SEGS; ++s) {
                const int srow = map[rr * SEGS + s];
                sp[s] = srow < 0 ? nullptr : x + ((size_t)b * src_rows_per_clip + srow) * Cp;
                dp[s] = srow < 0 ? nullptr : dx + ((size_t)b * src_rows_per_clip + srow) * Cp;
            }
            gp = dy + (size_t)row * (SEGS * Cp);
        } else {
            sp[0] = x + (size_t)row * Cp; dp[0] = dx + (size_t)row * Cp;
            gp = (MODE == 1) ? dy + ((size_t)b * dy_rows_per_clip + map[rr]) * Cp : dy + (size_t)row * Cp;
        }
        // the row (input and upstream gradient) is read ONCE into registers: the four passes below (mean, variance, the two projections, dx) walked it
        // four times through L1 before (2.2 TB/s of algorithmic bytes at C = 144)
        f32x4 xr[SEGS][MAXV], gr[SEGS][MAXV];
#pragma unroll
        for (int s = 0; s < SEGS; ++s)
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int v = sub + 16 * q;
                xr[s][q] = (v < V && sp[s]) ? ld4(sp[s] + 4 * v) : zero4();
                gr[s][q] = (v < V) ? ld4(gp + s * Cp + 4 * v) : zero4();
            }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < SEGS; ++s)
            if (sp[s])
#pragma unroll
                for (int q = 0; q < MAXV; ++q) {
                    const int v = sub + 16 * q;
                    if (v < V) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (4 * v + e < C) sum += xr[s][q][e];
                    }
                }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float mean = sum / (float)(SEGS * C);
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < SEGS; ++s)
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int v = sub + 16 * q;
                if (v < V) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (4 * v + e < C) { const float d = xr[s][q][e] - mean; var += d * d; }
                }
            }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) var += __shfl_xor(var, o, 16);
        const float rstd = 1.0f / sqrtf(var / (float)(SEGS * C) + eps);
        // c1 = mean(dy*gamma), c2 = mean(dy*gamma*xhat); dgamma += dy*xhat, dbeta += dy
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int v = sub + 16 * q;
                if (v < V) {
                    const f32x4 xv = xr[s][q];
                    const f32x4 g = gr[s][q], gm = ld4(gamma + s * Cp + 4 * v);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * v + e < C) {
                            const float xh = (xv[e] - mean) * rstd, t = g[e] * gm[e];
                            s1 += t; s2 += t * xh;
                            ag[s][q][e] += g[e] * xh; ab[s][q][e] += g[e];
                        }
                }
            }
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o, 16); s2 += __shfl_xor(s2, o, 16); }
        const float c1 = s1 / (float)(SEGS * C), c2 = s2 / (float)(SEGS * C);
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
            if (!dp[s]) continue;
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int v = sub + 16 * q;
                if (v >= V) continue;
                const f32x4 xv = xr[s][q];
                const f32x4 g = gr[s][q], gm = ld4(gamma + s * Cp + 4 * v);
                f32x4 o = (MODE != 2 && add) ? ld4(add + (size_t)row * Cp + 4 * v) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[e] - mean) * rstd;
                    o[e] = (4 * v + e < C) ? o[e] + rstd * (g[e] * gm[e] - c1 - xh * c2) : 0.f;
                }
                st4(dp[s] + 4 * v, o);
                // MODE 0 only: a second copy in window-slot order (token -> slot through slot_of), so that the projection's dW / dX GEMMs
                // read plain rows instead of gathering through the window map
                if (MODE == 0 && dx_slots) st4(dx_slots + ((size_t)b * slots_per_clip + slot_of[rr]) * Cp + 4 * v, o);
            }
        }
    }
    // workgroup reduction of the per-thread dgamma / dbeta accumulators: group 0 + 1 + ... + 15, fixed order
    const int RW = SEGS * Cp;
#pragma unroll
    for (int s = 0; s < SEGS; ++s) {
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
            const int v = sub + 16 * q;
            if (v < V) {
                st4(dyn + ((size_t)gl * 2 + 0) * RW + s * Cp + 4 * v, ag[s][q]);
                st4(dyn + ((size_t)gl * 2 + 1) * RW + s * Cp + 4 * v, ab[s][q]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * RW; i += 256) {
        float t = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += dyn[(size_t)g2 * 2 * RW + i];
        part[(size_t)blockIdx.x * 2 * RW + i] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Window attention core backward (attention.py:222-241), one wave per (window, head); the softmax rows are recomputed from q, k.
//   qkv  : [B*nW*16][ldq]  saved forward tensor (q already scaled)       dout : [B*nW*16][ldo]  gradient of the head-concatenated output
//   dqkv : [B*nW*16][ldq]  gradient w.r.t. the UNSCALED qkv linear output (q columns are multiplied by `scale` here)
//   dbias_part : [gridDim.x][nH][16][16]  per-workgroup sums of dS (relative-position bias gradient), reduced afterwards in fixed order
// grid = (window chunks, nH / HPW): a wave walks windows and, on each, its HPW heads one after the other (their dS sums stay in registers).  Heads that
// share the rows of a window in different workgroups re-fetch the same cache lines whenever they drift apart (measured: 3.0 TB/s of 1-pass traffic with one
// head per wave and the chip full, 4.9 TB/s with a third of the waves); with the heads of a window in ONE wave every line is fetched once.
// ------------------------------------------------------------------------------------------------
template <int STEPS, int HPW>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ dout,
                                                       float* __restrict__ dqkv, float* __restrict__ dbias_part, int total_windows, int nH,
                                                       int ldq, int ldo, int nWh, int nWw, int shifted, float scale) {
    constexpr int HDP = 4 * STEPS;
    constexpr int DT = (HDP + 15) / 16;
    constexpr bool STAGE = HDP <= 32;            // q / k / dO tiles of the window are kept in LDS for the column-wise second uses
    constexpr int TL = STAGE ? HDP + 1 : 1;
    __shared__ float tp[4][2][16][17];           // per wave: P and dS, to read them transposed
    __shared__ float tile[4][3][16][TL];         // per wave: Q, K, dO (row = token, column = head dim)
    __shared__ float wsum[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int h0 = blockIdx.y * HPW;             // this wave's heads: h0 .. h0 + HPW - 1, one after the other on the same window
    const int wave_global = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int kOff = nH * HDP, vOff = 2 * nH * HDP;
    f32x4 dbsum[HPW];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) dbsum[hh] = zero4();
    // the four operand rows of the NEXT (window, head) are fetched while the current one is processed (one exposed memory round trip per
    // window instead of three)
    float nq[STEPS], nk[STEPS], nv[STEPS], nd[STEPS];
    // element offsets as 32-bit running values (the launcher guarantees total_windows * 16 * ld < 2^32): the per-window 64-bit multiplies and
    // shifts of the address arithmetic were a third of this kernel's VALU instructions (31 v_lshl_add_u64 + 16 v_mul_lo_u32 per 20 MFMAs)
    const unsigned rowq = (unsigned)i * ldq + h0 * HDP, rowo = (unsigned)i * ldo + h0 * HDP;
    unsigned wq = (unsigned)wave_global * 16u * ldq, wo = (unsigned)wave_global * 16u * ldo;
    const unsigned stepq = (unsigned)nwaves * 16u * ldq, stepo = (unsigned)nwaves * 16u * ldo;
    auto fetch = [&](unsigned fq, unsigned fo, int hh) {
        const float* rowp = qkv + (fq + rowq + hh * HDP + STEPS * g);
        const float* dorow = dout + (fo + rowo + hh * HDP + STEPS * g);
#pragma unroll
        for (int r = 0; r < STEPS; ++r) { nq[r] = rowp[r]; nk[r] = rowp[kOff + r]; nv[r] = rowp[vOff + r]; nd[r] = dorow[r]; }
    };
    if (wave_global < total_windows) fetch(wq, wo, 0);
    for (int win = wave_global; win < total_windows; win += nwaves, wq += stepq, wo += stepo) {
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
        const int h = h0 + hh;
        const float* base = qkv + (wq + h * HDP);
        float kf[STEPS], qf[STEPS], vf[STEPS], df[STEPS];
#pragma unroll
        for (int r = 0; r < STEPS; ++r) { qf[r] = nq[r]; kf[r] = nk[r]; vf[r] = nv[r]; df[r] = nd[r]; }
        if (hh + 1 < HPW) fetch(wq, wo, hh + 1);
        else if (win + nwaves < total_windows) fetch(wq + stepq, wo + stepo, 0);
        if (STAGE) {
#pragma unroll
            for (int r = 0; r < STEPS; ++r) {
                tile[wave][0][i][(STEPS * g + r) % TL] = qf[r]; tile[wave][1][i][(STEPS * g + r) % TL] = kf[r]; tile[wave][2][i][(STEPS * g + r) % TL] = df[r];
            }
        }
        f32x4 s = zero4(), dp = zero4();
#pragma unroll
        for (int r = 0; r < STEPS; ++r) {
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[r], qf[r], s, 0, 0, 0);       // lane (i, g): S[i][4g + r']
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], df[r], dp, 0, 0, 0);     // lane (i, g): dP[i][4g + r'] = dO[i] . V[4g + r']
        }
        s += ld4(bias + ((size_t)h * 16 + i) * 16 + 4 * g);
        if (shifted) {
            const int wloc = win % (nWh * nWw);
            const int wh = wloc / nWw, ww = wloc - wh * nWw;
            const bool lastH = (wh == nWh - 1), lastW = (ww == nWw - 1);
            const int qh = i >> 2, qw = i & 3;
            const int labq = 3 * (lastH ? (qh < 2 ? 1 : 2) : 0) + (lastW ? (qw < 2 ? 1 : 2) : 0);
            const int labkh = 3 * (lastH ? (g < 2 ? 1 : 2) : 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int labk = labkh + (lastW ? (r < 2 ? 1 : 2) : 0);
                s[r] += (labk != labq) ? -100.0f : 0.0f;
            }
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        mx = max_groups(mx);                             // v_permlane16/32_swap butterflies (gemm_engine.h)
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = exp_fast(s[r] - mx);
        float den = (p[0] + p[1]) + (p[2] + p[3]);
        den = sum_groups(den);
        const float inv = 1.0f / den;
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] *= inv; dot += p[r] * dp[r]; }
        dot = sum_groups(dot);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = p[r] * (dp[r] - dot);
        dbsum[hh] += ds;
        // transposed P and dS through LDS: lane (j, g) then holds P[4g + r][j]
#pragma unroll
        for (int r = 0; r < 4; ++r) { tp[wave][0][i][4 * g + r] = p[r]; tp[wave][1][i][4 * g + r] = ds[r]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        f32x4 pt, dsT;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pt[r] = tp[wave][0][4 * g + r][i]; dsT[r] = tp[wave][1][4 * g + r][i]; }
        // dQ[i][d] = sum_j dS[i][j] K[j][d]  (same operand pattern as O = P V in the forward), scaled back through q*scale;
        // dV[j][d] = sum_i P[i][j] dO[i][d] ; dK[j][d] = sum_i dS[i][j] Q[i][d]   (lane (j, g): rows of key j)
        float* dqrow = dqkv + (wq + (unsigned)i * ldq + h * HDP);
        float* dkrow = dqrow + kOff;
        float* dvrow = dqrow + vOff;
        const float* dobase = dout + (wo + h * HDP);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int d = t * 16 + i;
            f32x4 oq = zero4(), ov = zero4(), ok = zero4();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float kk, dd, qq;
                if (STAGE) {
                    const int dc = d < HDP ? d : 0;
                    kk = tile[wave][1][4 * g + r][dc % TL]; dd = tile[wave][2][4 * g + r][dc % TL]; qq = tile[wave][0][4 * g + r][dc % TL];
                    if (d >= HDP) { kk = 0.f; dd = 0.f; qq = 0.f; }
                } else {
                    kk = (d < HDP) ? base[(size_t)(4 * g + r) * ldq + kOff + d] : 0.f;
                    dd = (d < HDP) ? dobase[(size_t)(4 * g + r) * ldo + d] : 0.f;
                    qq = (d < HDP) ? base[(size_t)(4 * g + r) * ldq + d] : 0.f;
                }
                oq = __builtin_amdgcn_mfma_f32_16x16x4f32(kk, ds[r], oq, 0, 0, 0);
                ov = __builtin_amdgcn_mfma_f32_16x16x4f32(dd, pt[r], ov, 0, 0, 0);
                ok = __builtin_amdgcn_mfma_f32_16x16x4f32(qq, dsT[r], ok, 0, 0, 0);
            }
            if (t * 16 + 4 * g < HDP) { st4(dqrow + t * 16 + 4 * g, oq * scale); st4(dvrow + t * 16 + 4 * g, ov); st4(dkrow + t * 16 + 4 * g, ok); }
        }
        __builtin_amdgcn_wave_barrier();             // the LDS tiles are rewritten by the next window
        if (h == 0 && 3 * nH * HDP < ldq)            // tail padding of the qkv width: keep exact zeros (K padding of the dX GEMM)
            for (int c = 3 * nH * HDP + g; c < ldq; c += 4) dqkv[wq + (unsigned)i * ldq + c] = 0.f;
        }
    }
    // relative-position bias gradient: sum the four waves in fixed order, one partial [16][16] per (workgroup, head)
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        if (hh) __syncthreads();
        st4(&wsum[wave][lane][0], dbsum[hh]);
        __syncthreads();
        if (wave == 0) {
            f32x4 t = ld4(&wsum[0][lane][0]);
            t += ld4(&wsum[1][lane][0]); t += ld4(&wsum[2][lane][0]); t += ld4(&wsum[3][lane][0]);
            st4(dbias_part + (((size_t)blockIdx.x * nH + h0 + hh) * 16 + i) * 16 + 4 * g, t);
        }
    }
}

// table gradient: dtable[idx][h] = sum over the (i, j) pairs of the 4x4 window with relative index idx of dbias[h][i][j], fixed order
static __global__ void bias_table_grad_kernel(const float* __restrict__ dbias, float* __restrict__ dtable, int nH) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 49 * nH) return;
    const int idx = e / nH, h = e - idx * nH;
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            if (((i >> 2) - (j >> 2) + 3) * 7 + ((i & 3) - (j & 3) + 3) == idx) s += dbias[((size_t)h * 16 + i) * 16 + j];
    dtable[e] = s;
}

// ------------------------------------------------------------------------------------------------
// small element-wise helpers
// ------------------------------------------------------------------------------------------------
static __global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st4(dst + 4 * i, ld4(dst + 4 * i) + ld4(src + 4 * i));
}
// src[0..n) -> a, src[n..2n) -> b
static __global__ void copy2_kernel(const float* __restrict__ src, float* __restrict__ a, float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = src[i];
    else if (i < 2 * n) b[i - n] = src[i];
}
static __global__ void add_tail_kernel(float* __restrict__ dst, const float* __restrict__ src, long long from, long long n) {
    const long long i = from + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
static __global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, long long per_row, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * g[i / per_row];
}
// derived layouts refreshed on the device from the flat fp32 parameter buffer: arena[i] = flat[map[i] - 1] (map 0 = structural zero, < 0 = computed elsewhere)
static __global__ void gather_params_kernel(const float* __restrict__ flat, const int* __restrict__ map, float* __restrict__ arena, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = map[i];
    if (c > 0) arena[i] = flat[c - 1];
    else if (c == 0) arena[i] = 0.f;
}
// gradient of the packed layouts back to the flat reference layout (one-to-one on the regions it is launched on)
// packed gradients -> flat reference layout, all primary layouts in ONE launch: seg[s] = (first element of segment s in the concatenation,
// arena offset of the segment); element i belongs to the last segment whose first element is <= i (binary search over ~350 entries)
static __global__ void scatter_grads_kernel(const float* __restrict__ garena, const int* __restrict__ map, float* __restrict__ gflat,
                                            const long long* __restrict__ seg, int nseg, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg[2 * mid] <= i) lo = mid; else hi = mid - 1; }
    const long long a = seg[2 * lo + 1] + (i - seg[2 * lo]);
    const int c = map[a];
    if (c > 0) gflat[c - 1] = garena[a];
}
// F.normalize of the codebooks + squared norms (codebook.py:31-36), same summation order as the host packer
static __global__ void codebook_normalize_kernel(const float* __restrict__ raw, float* __restrict__ cbn, float* __restrict__ c2, int rows, int d, int dt, int l2norm) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= rows) return;
    const float* r = raw + (size_t)k * dt;
    float ss = 0.f;
    for (int j = 0; j < d; ++j) ss += r[j] * r[j];
    const float den = l2norm ? fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
    float s2 = 0.f;
    for (int j = 0; j < dt; ++j) { const float v = j < d ? r[j] / den : 0.f; cbn[(size_t)k * dt + j] = v; s2 += v * v; }
    c2[k] = s2;
}

// ------------------------------------------------------------------------------------------------
// Product VQ in training mode (codebook.py:57-75, quantization.py:53-64).  One thread per (vector m, group g).
//   ze  [M][ldz]  projected vectors, group g at columns g*dt .. g*dt+d-1         codes[b*bstride + g*Tq + t]
//   zup [M][ldz]  what the up-projection sees: STE value z_e + (z_q - z_e), or with frozen codebooks (z_q_ste * 0 + z_e)
//   terms[g*M + m] = sum_j (z_q - z_e)^2 * loss_scale   (commitment == codebook loss numerically; zero when frozen)
// ------------------------------------------------------------------------------------------------
static __global__ void pvq_train_fwd_kernel(const float* __restrict__ ze, const long long* __restrict__ codes, long long bstride, const float* __restrict__ cbraw,
                                     float* __restrict__ zup, float* __restrict__ terms, int M, int G, int Ksz, int d, int dt, int ldz, int Tq,
                                     float loss_scale, int freeze) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * G) return;
    const int g = e / M, m = e - g * M;
    const int b = m / Tq, t = m - b * Tq;
    const long long code = codes[(size_t)b * bstride + (size_t)g * Tq + t];
    const float* q = cbraw + ((size_t)g * Ksz + (size_t)code) * dt;
    const float* z = ze + (size_t)m * ldz + g * dt;
    float* o = zup + (size_t)m * ldz + g * dt;
    float acc = 0.f;
    for (int j = 0; j < dt; ++j) {
        if (j < d) {
            const float df = q[j] - z[j];
            acc += df * df;
            const float ste = z[j] + df;                      // z_e + (z_q - z_e).detach()
            o[j] = freeze ? ste * 0.f + z[j] : ste;
        } else o[j] = 0.f;
    }
    terms[(size_t)g * M + m] = freeze ? 0.f : acc * loss_scale;
    if (g == 0) for (int c = G * dt; c < ldz; ++c) zup[(size_t)m * ldz + c] = 0.f;
}

// d z_e = d z_up (straight-through / frozen pass-through) + d(cm_loss) ;  gq = d(cb_loss) w.r.t. the selected code row (per vector)
static __global__ void pvq_train_bwd_kernel(const float* __restrict__ ze, const long long* __restrict__ codes, long long bstride, const float* __restrict__ cbraw,
                                     const float* __restrict__ dzup, const float* __restrict__ dcm, const float* __restrict__ dcb, float* __restrict__ dze,
                                     float* __restrict__ gq, int M, int G, int Ksz, int d, int dt, int ldz, int Tq, float loss_scale, int freeze) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * G) return;
    const int g = e / M, m = e - g * M;
    const int b = m / Tq, t = m - b * Tq;
    const long long code = codes[(size_t)b * bstride + (size_t)g * Tq + t];
    const float* q = cbraw + ((size_t)g * Ksz + (size_t)code) * dt;
    const float* z = ze + (size_t)m * ldz + g * dt;
    const float wm = (dcm && !freeze) ? 2.0f * loss_scale * dcm[b] : 0.f;
    const float wb = (dcb && !freeze) ? 2.0f * loss_scale * dcb[b] : 0.f;
    for (int j = 0; j < dt; ++j) {
        const size_t o = (size_t)m * ldz + g * dt + j;
        if (j < d) { const float df = z[j] - q[j]; dze[o] = dzup[o] + wm * df; gq[o] = -wb * df; }
        else { dze[o] = 0.f; gq[o] = 0.f; }
    }
    if (g == 0) for (int c = G * dt; c < ldz; ++c) dze[(size_t)m * ldz + c] = 0.f;
}

// embedding gradient without atomics.  One workgroup per (group, 16 codes): the group's codes are staged once in LDS as int16 (the
// per-element b / t division happens there, not once per code), then each of the 4 waves takes 4 codes in turn and scans the vectors 64 at a
// time.  Round 4: the vectors that chose the code are first COLLECTED - ballot + prefix rank into a wave-private LDS list, in increasing vector
// order - and the list is summed in batches of F = SL * 16 rows with all 16 row loads of a lane in flight together: the 64 lanes are SL = 64 / DTP
// slots of DTP dimensions, slot s adds the hits whose position in the code's hit order is = s (mod SL), in increasing order, and the slots are
// added in slot order at the end - a fixed order whatever the launch geometry.  Rounds 2-3 added the hit rows one dependent load (then four) at a
// time: with a skewed code usage (a few codes chosen by hundreds of vectors - every untrained codebook) that chain was the kernel, 200 us per stream.
constexpr int CBG_CODES = 16, CBG_MAXM = 24576, CBG_U = 16;
template <int DTP>
static __global__ __launch_bounds__(256) void codebook_grad_kernel(const long long* __restrict__ codes, long long bstride, const float* __restrict__ gq,
                                                            float* __restrict__ dcb, int M, int G, int Ksz, int dt, int ldz, int Tq) {
    constexpr int SL = 64 / DTP, F = SL * CBG_U;            // rows per batch (a multiple of SL: a hit's slot does not depend on the batch it lands in)
    __shared__ short lc[CBG_MAXM];
    __shared__ int hits[4][F + 64];                         // per wave: pending hit rows (vector indices), at most F - 1 left over + one window of 64
    const int blocks_per_group = (Ksz + CBG_CODES - 1) / CBG_CODES;
    const int g = blockIdx.x / blocks_per_group, k0 = (blockIdx.x - g * blocks_per_group) * CBG_CODES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane / DTP, j = lane - slot * DTP;
    int* hl = hits[wave];
    float acc[CBG_CODES / 4];
#pragma unroll
    for (int c = 0; c < CBG_CODES / 4; ++c) acc[c] = 0.f;
    const float* gcol = gq + g * dt + j;
    auto flush = [&](float& a, int n) {                      // adds the first n (<= F) list entries: lane (slot, j) takes entries slot, slot + SL, ...
        float v[CBG_U];
#pragma unroll
        for (int u = 0; u < CBG_U; ++u) {
            const int i = u * SL + slot;
            v[u] = (i < n && j < dt) ? gcol[(size_t)hl[i] * ldz] : 0.0f;      // a missing entry adds 0.0f: the sum is bitwise unchanged
        }
#pragma unroll
        for (int u = 0; u < CBG_U; ++u) a += v[u];
    };
    for (int mbase = 0; mbase < M; mbase += CBG_MAXM) {                      // M beyond the LDS image: walk it in pieces, still in vector order
        const int mcnt = min(CBG_MAXM, M - mbase);
        __syncthreads();
        for (int i = threadIdx.x; i < mcnt; i += 256) {
            const int m = mbase + i; const int b = m / Tq, t = m - b * Tq;
            lc[i] = (short)codes[(size_t)b * bstride + (size_t)g * Tq + t];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CBG_CODES / 4; ++c) {
            const int k = k0 + wave * (CBG_CODES / 4) + c;
            int cnt = 0;                                                     // wave-uniform
            for (int m0 = 0; m0 < mcnt; m0 += 64) {
                const bool hit = (m0 + lane < mcnt) && lc[m0 + lane] == (short)k;
                const unsigned long long mask = __ballot(hit);
                if (!mask) continue;
                if (hit) hl[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = mbase + m0 + lane;
                cnt += __popcll(mask);
                while (cnt >= F) {                                           // one batch; the (< 64) entries behind it move to the front
                    flush(acc[c], F);
                    const int rest = cnt - F;
                    const int mv = lane < rest ? hl[F + lane] : 0;
                    if (lane < rest) hl[lane] = mv;
                    cnt = rest;
                }
            }
            if (cnt) flush(acc[c], cnt);
        }
    }
#pragma unroll
    for (int c = 0; c < CBG_CODES / 4; ++c) {
        float a = acc[c], tot = a;
#pragma unroll
        for (int sidx = 1; sidx < SL; ++sidx) tot += __shfl(a, sidx * DTP + j, 64);   // slot order 0, 1, ...: every lane computes the same chain
        const int k = k0 + wave * (CBG_CODES / 4) + c;
        if (k < Ksz && slot == 0 && j < dt) dcb[((size_t)g * Ksz + k) * dt + j] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// inverse-STFT backward: gradient of the windowed frames from the waveform gradient (istft_ola_kernel transposed)
// ------------------------------------------------------------------------------------------------
static __global__ void istft_ola_bwd_kernel(const float* __restrict__ dwave, const float* __restrict__ win2, float* __restrict__ dframes, int B, int T,
                                     int ldf, int win, int hop, int left, int half, int out_len) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * T * ldf) return;
    const int j = (int)(idx % ldf); const long long bt = idx / ldf; const int t = (int)(bt % T), b = (int)(bt / T);
    float v = 0.f;
    if (j < win) {
        const int p = t * hop + j;                       // position relative to frame 0's window support
        const int s = p - (half - left);
        if (s >= 0 && s < out_len) {
            int t_hi = p / hop; if (t_hi > T - 1) t_hi = T - 1;
            float env = 0.f;
            for (int tt = t_hi; tt >= 0; --tt) { const int jj = p - tt * hop; if (jj >= win) break; env += win2[jj]; }
            v = dwave[(size_t)b * out_len + s] / env;
        }
    }
    dframes[idx] = v;
}

// conv3x3 dX on the fine time-major map: d_in[(b,t,f)][ci] = sum_{tap,oc} g[(b, t-(kw-1))][oc*Fp + f-(kh-1)] * w[oc][tap=(kw*3+kh)][ci]
// (w is the packed [16][9*Cp] matrix of the forward implicit GEMM: tap order (t0 over time = kw, t1 over freq = kh))
static __global__ void conv3_dx_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, int B, int T, int F, int Fp, int Cp,
                                int in_dim) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (b,t,f, 4 channels)
    const int V = Cp / 4;
    if (idx >= (long long)B * T * F * V) return;
    const int c4 = (int)(idx % V); long long r = idx / V;
    const int f = (int)(r % F); r /= F; const int t = (int)(r % T), b = (int)(r / T);
    f32x4 acc = zero4();
    for (int a = 0; a < 3; ++a) {               // a: time tap (t0), forward reads in(t + a - 1)
        const int ts = t - (a - 1);
        if (ts < 0 || ts >= T) continue;
        for (int bq = 0; bq < 3; ++bq) {        // freq tap (t1)
            const int fs = f - (bq - 1);
            if (fs < 0 || fs >= F) continue;
            for (int oc = 0; oc < in_dim; ++oc) {
                const float gv = g[((size_t)b * T + ts) * (in_dim * Fp) + oc * Fp + fs];
                acc += ld4(w + (size_t)oc * 9 * Cp + (size_t)(a * 3 + bq) * Cp + 4 * c4) * gv;
            }
        }
    }
    st4(dx + ((((size_t)b * T + t) * F + f) * Cp) + 4 * c4, acc);
}

// ------------------------------------------------------------------------------------------------
// De-embedding backward through its low-rank structure.  The gradient at the conv5x5 output, dY1[pix][(q, cc)] (270 channels), is
//     sum_{oc, a, b} W2[oc][cc][a][b] * g_pad[fine(pix, q) - (a-1, b-1)][oc]          (conv3x3 transposed; g = d loss / d spectrum),
// i.e. a fixed 45 x 18 matrix applied to the 18 spectrum-gradient values around each fine position.  So instead of running the two
// 270-channel contractions of conv5x5's dW and dX on dY1, they run on P[pix][q][j = (oc, a, b)] (6 x 18 -> 6 x 20 = 120 columns, padded
// to 128) and the 45 x 18 matrix is folded into the (tiny) weight side: 2.3x fewer FLOPs, exact including the zero-padded borders.
//   P      [B*H*W][128]   column q*20 + j, j = oc*9 + a*3 + b < 18 (a: frequency tap, b: time tap), zero elsewhere
//   g      frame-major spectrum gradient [(b, t)][oc*Fp + f]
// ------------------------------------------------------------------------------------------------
constexpr int DEP_J = 20, DEP_LD = 128;
// Round 4: the 6 x 18 columns of P are COPIES of only in_dim * (pf + 2) * (pt + 2) = 40 distinct spectrum-gradient values per coarse pixel (its pf x pt
// fine block plus a one-pixel halo), so the two big contractions run on D[pix][slot] (40 -> 48 columns instead of 128: 2.7x fewer FLOPs again) and
// the copy pattern moves to the weight side, exact including the zero-padded borders (a value outside the fine map is 0 in D as it was in every
// P column that referenced it).  slot(q = (s1, s2), j = (oc, a, b)) = oc * NS + (s1 - a + 2) * (pt + 2) + (s2 - b + 2), NS = (pf + 2)(pt + 2).
constexpr int DEP_LD2 = 48;
__device__ __forceinline__ int dep_row(int q, int j, int pf, int pt, bool slots) {
    if (!slots) return q * DEP_J + j;
    const int oc = j / 9, ab = j - oc * 9, a = ab / 3, bq = ab - a * 3;
    const int s1 = q / pt, s2 = q - s1 * pt;
    return oc * (pf + 2) * (pt + 2) + (s1 - a + 2) * (pt + 2) + (s2 - bq + 2);
}
static __global__ void deembed_d_kernel(const float* __restrict__ g, float* __restrict__ D, int B, int H, int W, int pf, int pt, int in_dim, int Fp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * H * W * DEP_LD2) return;
    const int col = (int)(idx % DEP_LD2); long long pix = idx / DEP_LD2;
    const int w = (int)(pix % W); pix /= W; const int h = (int)(pix % H), b = (int)(pix / H);
    const int NS = (pf + 2) * (pt + 2);
    float v = 0.f;
    if (col < in_dim * NS) {
        const int oc = col / NS, r = col - oc * NS, rf = r / (pt + 2), rt = r - rf * (pt + 2);
        const int f = pf * h + rf - 1, t = pt * w + rt - 1;
        if (f >= 0 && f < pf * H && t >= 0 && t < pt * W) v = g[((size_t)b * (pt * W) + t) * (in_dim * Fp) + oc * Fp + f];
    }
    D[idx] = v;
}
static __global__ void deembed_p_kernel(const float* __restrict__ g, float* __restrict__ P, int B, int H, int W, int pf, int pt, int in_dim, int Fp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * H * W * DEP_LD) return;
    const int col = (int)(idx % DEP_LD); long long pix = idx / DEP_LD;
    const int w = (int)(pix % W); pix /= W; const int h = (int)(pix % H), b = (int)(pix / H);
    const int q = col / DEP_J, j = col - q * DEP_J;
    float v = 0.f;
    if (q < pf * pt && j < in_dim * 9) {
        const int oc = j / 9, ab = j - oc * 9, a = ab / 3, bq = ab - a * 3;
        const int s1 = q / pt, s2 = q - s1 * pt;
        const int f = pf * h + s1 - (a - 1), t = pt * w + s2 - (bq - 1);
        if (f >= 0 && f < pf * H && t >= 0 && t < pt * W) v = g[((size_t)b * (pt * W) + t) * (in_dim * Fp) + oc * Fp + f];
    }
    P[idx] = v;
}
// dW1[(q*Cp + cc)][k] = sum_j W2[oc][cc][a][b] * R[q*20 + j][k] ; w2 is the packed conv3x3 matrix [16][9*Cp], tap order (b*3 + a)
static __global__ void deembed_fold_dw_kernel(const float* __restrict__ R, const float* __restrict__ w2, float* __restrict__ dW1, int Q, int C, int Cp, int K,
                                       int in_dim, int pf = 0, int pt = 1, bool slots = false) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Q * Cp * K) return;
    const int k = (int)(idx % K); const int r = (int)(idx / K); const int q = r / Cp, cc = r - q * Cp;
    float s = 0.f;
    if (cc < C)
        for (int j = 0; j < in_dim * 9; ++j) {
            const int oc = j / 9, ab = j - oc * 9, a = ab / 3, bq = ab - a * 3;
            s += w2[(size_t)oc * 9 * Cp + (size_t)(bq * 3 + a) * Cp + cc] * R[(size_t)dep_row(q, j, pf, pt, slots) * K + k];
        }
    dW1[idx] = s;
}
// conv3x3 weight gradient from the Q diagonal blocks X[q][j][cc] = sum_pix P[pix][q*20 + j] * Y1[pix][q*Cp + cc]:
//   dW2[oc][(b*3 + a)*Cp + cc] = sum_q X[q][j][cc],  j = oc*9 + a*3 + b ;   db2[oc] = sum_q Rb[q*20 + oc*9 + 4]  (centre tap: no border exclusion)
static __global__ void deembed_fold_dw2_kernel(const float* __restrict__ X, const float* __restrict__ Rb, float* __restrict__ dW2, float* __restrict__ db2, int Q,
                                        int C, int Cp, int in_dim, int pf = 0, int pt = 1, bool slots = false) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = in_dim * 9 * Cp;
    if (idx < n) {
        const int cc = idx % Cp; const int t = idx / Cp; const int tap = t % 9, oc = t / 9;      // tap = b*3 + a (packed order)
        const int bq = tap / 3, a = tap - bq * 3;
        float s = 0.f;
        if (cc < C)
            for (int q = 0; q < Q; ++q) s += X[((size_t)q * DEP_J + oc * 9 + a * 3 + bq) * Cp + cc];
        dW2[idx] = s;
    } else if (idx < n + in_dim) {
        const int oc = idx - n;
        float s = 0.f;
        for (int q = 0; q < Q; ++q) s += Rb[dep_row(q, oc * 9 + 4, pf, pt, slots)];
        db2[oc] = s;
    }
}
// effective dX weights: Weff[ci][(t0*5 + t1)*128 + q*20 + j] = sum_cc W2[oc][cc][a][b] * W1[(q, cc)][ci][kh = 4 - t0][kw = 4 - t1]
// (taps flipped so that the forward implicit-GEMM loader ConvA serves the transposed convolution); w1 is the packed conv5x5 matrix
// [Q*Cp][25*Cp] with k = (kh*5 + kw)*Cp + ci
static __global__ void deembed_weff_kernel(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ weff, int Q, int C, int Cp, int in_dim) {
    const int KE = 25 * DEP_LD;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cp * KE) return;
    const int k = (int)(idx % KE), ci = (int)(idx / KE);
    const int tap = k / DEP_LD, col = k - tap * DEP_LD; const int q = col / DEP_J, j = col - q * DEP_J;
    float s = 0.f;
    if (ci < C && q < Q && j < in_dim * 9) {
        const int t0 = tap / 5, t1 = tap - t0 * 5, kh = 4 - t0, kw = 4 - t1;
        const int oc = j / 9, ab = j - oc * 9, a = ab / 3, bq = ab - a * 3;
        for (int cc = 0; cc < C; ++cc)
            s += w2[(size_t)oc * 9 * Cp + (size_t)(bq * 3 + a) * Cp + cc] * w1[(size_t)(q * Cp + cc) * 25 * Cp + (size_t)(kh * 5 + kw) * Cp + ci];
    }
    weff[idx] = s;
}

// conv3x3 weight gradient WITHOUT the saved conv5x5 output (round 4): Y1[pix][(q, cc)] = b1[(q, cc)] + sum_k W1[(q, cc)][k] * patch5x5(x0)[pix][k], so
//   X[q][j][cc] = sum_pix P[pix][q*20 + j] * Y1[pix][(q, cc)] = sum_k R[q*20 + j][k] * W1[(q, cc)][k] + Rb[q*20 + j] * b1[(q, cc)]
// with R = P^T . patches (the product the conv5x5 weight gradient already needs) and Rb = the column sums of P: the fine map never has to exist.
// One wave per (q, j) row: lanes stride k, 64-lane butterfly, every cc.  w1 = packed conv5x5 matrix [Q*Cp][K], b1 [Q*Cp].
static __global__ __launch_bounds__(256) void deembed_x_from_r_kernel(const float* __restrict__ R, const float* __restrict__ Rb, const float* __restrict__ w1,
                                                                       const float* __restrict__ b1, float* __restrict__ X, int Q, int C, int Cp, int K, int nj,
                                                                       int pf = 0, int pt = 1, bool slots = false) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= Q * nj * Cp) return;
    const int cc = wave % Cp; const int t = wave / Cp; const int j = t % nj, q = t / nj;
    float s = 0.f;
    if (cc < C) {
        const int row = dep_row(q, j, pf, pt, slots);
        const float* r = R + (size_t)row * K;
        const float* w = w1 + (size_t)(q * Cp + cc) * K;
        for (int k = lane; k < K; k += 64) s = fmaf(r[k], w[k], s);
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft);
        s += Rb[row] * b1[q * Cp + cc];
    }
    if (lane == 0) X[((size_t)q * DEP_J + j) * Cp + cc] = s;
}

// ------------------------------------------------------------------------------------------------
// Composed de-embedding ON THE DEVICE (round 4): conv3x3 o pixel_shuffle o conv5x5 folded into the 7x7 / 12-output map of the inference path
// (escx_api.cpp, "composed de-embedding"), from the CURRENT packed convolution weights, in fp64 and in the host fold's summation order - every
// element is the same sum of the same double products (contraction off: the host compiler does not fuse them either), so the fp32 results are the
// host's bit for bit.  With it the training forward runs the folded kernels (11x fewer FLOPs, no 270-channel map on the tape) and a model that
// has taken optimiser steps on the device needs no host round trip before it decodes again.
//   w1 [Q*Cp][25*Cp] (k = (kh*5 + kw)*Cp + ci), b1 [Q*Cp], w2 [16][9*Cp] (k = (kw*3 + kh)*Cp + ci), b2 [16]
//   outputs: all 16 border variants wv [16][NO][49*C], bv [16][NO]; the interior variant again as GEMM rows wc [16][49*Cp], bc [16] and as MFMA
//   fragments wh [tap][Cp/16][64][4] (lane (n = l & 15, g = l >> 4), channel 16kk + 4g + r)
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void deembed_compose_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                                      const float* __restrict__ b2, float* __restrict__ wv, float* __restrict__ bv,
                                                                      float* __restrict__ wc, float* __restrict__ bc, float* __restrict__ wh,
                                                                      int C, int Cp, int pf, int pt, int in_dim) {
#pragma clang fp contract(off)
    const int Q = pf * pt, NO = in_dim * Q, Kc = 49 * C, K1 = 25 * Cp, K2 = 9 * Cp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_w = (long long)16 * NO * Kc;
    if (idx >= n_w + 16 * NO) return;
    auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    const bool is_bias = idx >= n_w;
    int v, n, tap = 0, ci = 0;
    if (is_bias) { const int r = (int)(idx - n_w); v = r / NO; n = r - v * NO; }
    else { long long r = idx; ci = (int)(r % C); r /= C; tap = (int)(r % 49); r /= 49; n = (int)(r % NO); v = (int)(r / NO); }
    const int eh = v >> 2, ew = v & 3;
    const int co = n / Q, qq = n - co * Q, s1 = qq / pt, s2 = qq - s1 * pt;
    const int dh = tap / 7 - 3, dw = tap % 7 - 3;
    double acc = is_bias ? (double)b2[co] : 0.0;
    for (int a = 0; a < 3; ++a) for (int bq = 0; bq < 3; ++bq) {
        const int dh0 = fdiv(s1 + a - 1, pf), s1n = s1 + a - 1 - dh0 * pf;
        const int dw0 = fdiv(s2 + bq - 1, pt), s2n = s2 + bq - 1 - dw0 * pt;
        if ((dh0 < 0 && (eh & 1)) || (dh0 > 0 && (eh & 2)) || (dw0 < 0 && (ew & 1)) || (dw0 > 0 && (ew & 2))) continue;   // fine neighbour outside the map
        const int qn = s1n * pt + s2n;
        const int kh = dh - dh0 + 2, kw = dw - dw0 + 2;
        if (!is_bias && (kh < 0 || kh >= 5 || kw < 0 || kw >= 5)) continue;
        for (int cc = 0; cc < C; ++cc) {
            const double w2v = (double)w2[(size_t)co * K2 + (size_t)(bq * 3 + a) * Cp + cc];
            if (is_bias) acc += w2v * b1[qn * Cp + cc];
            else acc += w2v * w1[(size_t)(qn * Cp + cc) * K1 + (size_t)(kh * 5 + kw) * Cp + ci];
        }
    }
    const float r = (float)acc;
    if (is_bias) { bv[v * NO + n] = r; if (v == 0) bc[n] = r; return; }
    wv[idx] = r;
    if (v == 0) {
        wc[(size_t)n * 49 * Cp + (size_t)tap * Cp + ci] = r;
        if (NO <= 16) wh[((size_t)(tap * (Cp / 16) + ci / 16) * 64 + (n + 16 * ((ci % 16) / 4))) * 4 + (ci % 4)] = r;
    }
}

// effective dX weights over the distinct-value columns: Weff2[ci][(t0*5 + t1)*48 + slot] = sum over the (q, a, b) that reference the slot of
// sum_cc W2[oc][cc][a][b] * W1[(q, cc)][ci][4 - t0][4 - t1]   (slot = (oc, rf, rt): s1 = rf + a - 2, s2 = rt + b - 2)
static __global__ void deembed_weff_slots_kernel(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ weff, int C, int Cp, int pf, int pt,
                                                 int in_dim) {
    const int KE = 25 * DEP_LD2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cp * KE) return;
    const int k = (int)(idx % KE), ci = (int)(idx / KE);
    const int tap = k / DEP_LD2, col = k - tap * DEP_LD2;
    const int NS = (pf + 2) * (pt + 2);
    float s = 0.f;
    if (ci < C && col < in_dim * NS) {
        const int oc = col / NS, r = col - oc * NS, rf = r / (pt + 2), rt = r - rf * (pt + 2);
        const int t0 = tap / 5, t1 = tap - t0 * 5, kh = 4 - t0, kw = 4 - t1;
        for (int a = 0; a < 3; ++a) for (int bq = 0; bq < 3; ++bq) {
            const int s1 = rf + a - 2, s2 = rt + bq - 2;
            if (s1 < 0 || s1 >= pf || s2 < 0 || s2 >= pt) continue;
            const int q = s1 * pt + s2;
            for (int cc = 0; cc < C; ++cc)
                s += w2[(size_t)oc * 9 * Cp + (size_t)(bq * 3 + a) * Cp + cc] * w1[(size_t)(q * Cp + cc) * 25 * Cp + (size_t)(kh * 5 + kw) * Cp + ci];
        }
    }
    weff[idx] = s;
}

// ------------------------------------------------------------------------------------------------
// ComplexSTFTLoss with power-law compression (generator_loss.py:12-35): per clip mean over (2, F, T) of (pl(raw) - pl(recon))^2,
// pl(x) = sign(x) (|x| + 1e-10)^0.3.  Spectra are frame-major (B, T, in_dim, F).  part[b][block] partial sums; unit gradient optional.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float power_law(float x) { return (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) * powf(fabsf(x) + 1e-10f, 0.3f); }
static __global__ __launch_bounds__(256) void stft_loss_kernel(const float* __restrict__ raw, const float* __restrict__ rec, float* __restrict__ part,
                                                        float* __restrict__ grad, long long per_clip, int blocks_per_clip, float inv_n) {
    const int b = blockIdx.y;
    __shared__ float red[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (long long)blocks_per_clip * 256) {
        const float a = raw[(size_t)b * per_clip + i], r = rec[(size_t)b * per_clip + i];
        const float df = power_law(a) - power_law(r);
        acc += df * df;
        if (grad) {
            const float dpl = (r != 0.f) ? 0.3f * powf(fabsf(r) + 1e-10f, -0.7f) : 0.f;       // sign(r)^2 = 1 away from 0; sign(0) = 0 kills the term
            grad[(size_t)b * per_clip + i] = -2.0f * df * dpl * inv_n;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[(size_t)b * blocks_per_clip + blockIdx.x] = red[0] * inv_n;
}
// out[b] (+)= sum_k part[b][k], fixed order
static __global__ void row_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out, int rows, int accumulate, float scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= rows) return;
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += part[(size_t)b * n + k];
    out[b] = (accumulate ? out[b] : 0.f) + s * scale;
}

// ------------------------------------------------------------------------------------------------
// Mel loss pieces (generator_loss.py:37-74).  spec: [B*T][2*Fp] DFT rows (re | im) of raw and recon; mag = |S|.
// ------------------------------------------------------------------------------------------------
static __global__ void complex_mag_kernel(const float* __restrict__ spec, float* __restrict__ mag, long long rows, int F, int Fp, int Fq) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * Fq) return;
    const long long r = idx / Fq; const int f = (int)(idx - r * Fq);
    float v = 0.f;
    if (f < F) { const float re = spec[r * 2 * Fp + f], im = spec[r * 2 * Fp + Fp + f]; v = sqrtf(re * re + im * im); }
    mag[idx] = v;
}
// per-clip mel terms: |x - y| / n  +  |log10(clamp(x)^2) - log10(clamp(y)^2)| / n ; gradient w.r.t. y (the reconstruction's mel)
static __global__ __launch_bounds__(256) void mel_l1_kernel(const float* __restrict__ xm, const float* __restrict__ ym, float* __restrict__ part,
                                                     float* __restrict__ gy, int rows_per_clip, int n_mels, int ldm, int blocks_per_clip, float inv_n,
                                                     float clamp_eps) {
    const int b = blockIdx.y;
    __shared__ float red[256];
    const long long per = (long long)rows_per_clip * ldm;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)blocks_per_clip * 256) {
        const int c = (int)(i % ldm);
        const size_t o = (size_t)b * per + i;
        float gv = 0.f;
        if (c < n_mels) {
            const float x = xm[o], y = ym[o];
            const float xc = fmaxf(x, clamp_eps), yc = fmaxf(y, clamp_eps);
            const float lx = log10f(xc * xc), ly = log10f(yc * yc);
            acc += fabsf(x - y) + fabsf(lx - ly);
            const float s1 = (y > x) ? 1.f : (y < x ? -1.f : 0.f);
            const float s2 = (ly > lx) ? 1.f : (ly < lx ? -1.f : 0.f);
            gv = s1 * inv_n + ((y >= clamp_eps) ? s2 * inv_n * 0.86858896380650365530f / yc : 0.f);      // d log10(y^2)/dy = 2 / (y ln 10)
        }
        if (gy) gy[o] = gv;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[(size_t)b * blocks_per_clip + blockIdx.x] = red[0] * inv_n;
}
// d spec = d mag * S / |S|   (|.| of a complex number; zero gradient at S = 0)
static __global__ void complex_mag_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ dmag, float* __restrict__ dspec, long long rows, int F,
                                       int Fp, int Fq) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * Fp) return;
    const long long r = idx / Fp; const int f = (int)(idx - r * Fp);
    float gre = 0.f, gim = 0.f;
    if (f < F) {
        const float re = spec[r * 2 * Fp + f], im = spec[r * 2 * Fp + Fp + f];
        const float m = sqrtf(re * re + im * im);
        if (m > 0.f) { const float gg = dmag[r * Fq + f] / m; gre = gg * re; gim = gg * im; }
    }
    dspec[r * 2 * Fp + f] = gre; dspec[r * 2 * Fp + Fp + f] = gim;
}
// framing backward with reflect padding (torch.stft center=True): dwave[b][j] (+)= sum of dframes over every (frame, tap) that read sample j,
// including the mirrored reads of the padded ends.  padded position i = t*hop + k, sample = reflect(i - pad).
static __global__ void frames_bwd_kernel(const float* __restrict__ dframes, float* __restrict__ dwave, int B, int L, int T, int hop, int n_fft, int ldf,
                                  int accumulate, int pad_left = -1) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L) return;
    const int b = (int)(idx / L), j = (int)(idx - (long long)b * L);
    const int pad = pad_left >= 0 ? pad_left : n_fft / 2;      // samples of reflect padding in front of sample 0 (the right side is as long as the frames reach)
    auto gather = [&](int i) {                  // sum over frames covering padded position i
        float s = 0.f;
        if (i < 0) return s;
        int t_hi = i / hop; if (t_hi > T - 1) t_hi = T - 1;
        for (int t = t_hi; t >= 0; --t) { const int k = i - t * hop; if (k >= n_fft) break; s += dframes[((size_t)b * T + t) * ldf + k]; }
        return s;
    };
    float s = gather(j + pad);
    if (j >= 1 && j <= pad) s += gather(pad - j);                              // left mirror: padded i < pad reads sample pad - i
    if (j <= L - 2) s += gather(pad + 2 * (L - 1) - j);                        // right mirror: padded i - pad >= L reads 2(L-1) - (i - pad); gather() is 0 past the last frame
    dwave[idx] = (accumulate ? dwave[idx] : 0.f) + s;
}

// ------------------------------------------------------------------------------------------------
// optimiser: global-norm clipping + AdamW on flat buffers (trainer_no_adv.py:116-117; torch.optim.AdamW semantics)
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += x[i] * x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// norm_out[0] = sqrt(sum part) ; norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
static __global__ void clip_coef_kernel(const float* __restrict__ part, int n, float max_norm, float* __restrict__ norm_out) {
    if (threadIdx.x || blockIdx.x) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    const float nrm = sqrtf(s);
    norm_out[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    norm_out[1] = c < 1.0f ? c : 1.0f;
}
static __global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                             const float* __restrict__ coef, float decay, float beta1, float omb1, float beta2, float omb2, float eps, float step_size,
                             float bc2_sqrt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * (coef ? coef[1] : 1.0f);
    float pi = p[i] * decay;                                  // param.mul_(1 - lr * weight_decay)
    const float mi = beta1 * m[i] + omb1 * gi;                // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = beta2 * v[i] + omb2 * gi * gi;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi; v[i] = vi;
    pi -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);      // param.addcdiv_(exp_avg, sqrt(exp_avg_sq) / sqrt(bc2) + eps, value = -lr / bc1)
    p[i] = pi;
}

}  // namespace escx
