// Fused shifted-window attention half of a Swin block for gfx950 (attention.py:129-176, 215-244):
//
//   x <- x + proj( softmax( (q*scale) k^T + rel_pos_bias [+ shift mask] ) v ),  q,k,v = qkv(LN(x))  per 4x4 window
//
// One wave owns TMW windows (16 tokens each) end to end, entirely in registers:
//   1. gathers its window's tokens through the (pad, roll, partition) index map straight into the MFMA
//      operand layout and LayerNorms them in registers; padded slots become zero rows AFTER the norm
//      (attention.py:135-143), and still take part as keys/values with k = v = bias (attention.py:150-151);
//   2. per head group: Q and K tiles with the weights as the row operand -> lane (token, 4 dims) == the
//      operand layout of S^T = K.Q^T, so scores come straight from the accumulators; V with the operands
//      swapped -> lane (dim, 4 tokens) == the row-operand layout of O^T = V^T.P^T; the softmaxed P tile is
//      already the column operand.  O^T leaves lane (token, 4 dims) == operand layout of the projection;
//   3. projection accumulates over head groups; epilogue adds bias + shortcut and scatters 16 B per lane
//      back through the index map (window reverse + un-roll + crop).
// q, k, v, the 16x16 scores and the attention output never touch memory (the unfused path writes/reads
// 4x the activation for them).  Head dims map to 16-row MFMA tiles in three ways (MODE):
//   0: head_dim <= 16, one head per tile      (15, 12, 16 in ESC-Base)
//   1: head_dim <= 8, two heads per tile      (8, 6): scores via k-slot masking, outputs by lane select
//   2: head_dim <= 32, one head over two tiles (24)
// The NW waves of a workgroup stream the weights (host-packed in fragment order) through a double-buffered
// LDS ring filled by global_load_lds_dwordx4, UT tiles per stage.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gemm_engine.h"
#include "gemm_bf16.h"
#include "split_terms.h"

namespace escx {

// ---- split-operand (3 x bf16) form of the Q / K / V projections (round 5, see fused_mlp_x3.h for the arithmetic) ----
// The K = C contractions of a head group's Q, K and V tiles run on v_mfma_f32_16x16x32_bf16 with every fp32 operand split exactly into three bf16
// terms (six cross products, fp32 accumulation); their outputs land in the same accumulator layout as the fp32 MFMA's, so the 16 x 16 attention
// itself (scores, softmax, P.V) and the output projection (K = 16 per head group) stay on the fp32 MFMA unchanged.
// Weight stream of the X3 instantiations: [group][tile][TF][64 lanes][16 B], TF = max(3 KS, KK) fragments per tile; a Q / K / V tile holds its
// 3 KS split fragments (K-step s, term i) -> lane (row, g): term i of W[row][32 s + 8 g + e]; a projection tile holds its KK fp32 fragments as before.
constexpr int attn_x3_ks(int CP) { return (CP + 31) / 32; }
constexpr int attn_x3_tf(int CP) { return 3 * attn_x3_ks(CP) > CP / 16 ? 3 * attn_x3_ks(CP) : CP / 16; }
__device__ __forceinline__ void attn_split3(const float (&v)[8], bf16x8& t0, bf16x8& t1, bf16x8& t2) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 a1 = (__bf16)v[e];
        const float r1 = v[e] - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;
        t0[e] = a1; t1[e] = a2; t2[e] = (__bf16)r2;
    }
}
// Pair-order stream of the X3P instantiations, from the fp32 fragment stream ([group][Q K V P][KK][64] float4): per pair of groups eight tiles of TF fragments,
// [Q0 K0 V0 Q1 K1 V1 P_lo P_hi]; the projection halves hold, per output tile `to`, three fragments (term i) -> lane (channel n, g): term i of
// (e < 4 ? Wp_group0[16 to + n][4 g + e] : Wp_group1[16 to + n][4 g + e - 4]).  One thread per (pair, tile, fragment triple, lane).
__global__ __launch_bounds__(256) void attn_x3p_pack_kernel(const f32x4* __restrict__ waf, bf16x8* __restrict__ out, int n_pairs, int KK, int KS, int TF) {
    const int H = (KK + 1) / 2, per_tile = (KS > H ? KS : H) * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n_pairs * 8 * per_tile) return;
    const int tile8 = (int)(idx / per_tile), r = (int)(idx - (long long)tile8 * per_tile), f = r >> 6, lane = r & 63;
    const int pair = tile8 >> 3, ti = tile8 & 7;
    const int l15 = lane & 15, g = lane >> 4;
    bf16x8* dst = out + ((size_t)pair * 8 + ti) * TF * 64;
    float v[8];
    int fslot;
    if (ti < 6) {
        if (f >= KS) return;
        const f32x4* src = waf + ((size_t)(2 * pair + ti / 3) * 4 + ti % 3) * KK * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 2 * f + (g >> 1), lgs = 2 * (g & 1) + (e >> 2);
            v[e] = kk < KK ? src[kk * 64 + 16 * lgs + l15][e & 3] : 0.f;
        }
        fslot = f;
    } else {
        const int to = (ti == 6 ? 0 : H) + f;
        if (f >= (ti == 6 ? H : KK - H)) return;
        const f32x4* p0 = waf + ((size_t)(2 * pair) * 4 + 3) * KK * 64, *p1 = waf + ((size_t)(2 * pair + 1) * 4 + 3) * KK * 64;
        const f32x4 a0 = p0[to * 64 + lane], a1 = p1[to * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
        fslot = f;
    }
    bf16x8 t0, t1, t2;
    attn_split3(v, t0, t1, t2);
    dst[(fslot * 3 + 0) * 64 + lane] = t0; dst[(fslot * 3 + 1) * 64 + lane] = t1; dst[(fslot * 3 + 2) * 64 + lane] = t2;
}
// builds the X3 stream from the fp32 fragment stream (BlockW::waf): one thread per (group, tile, K-step or projection fragment, lane)
// NT = 2 (split_terms.h): two fp16 terms per weight, scaled by the power of two of the block's max |w| - the stream then ends with two 16-byte slots,
// [0] bits of max |w| (absmax_bits_kernel over the fp32 fragment stream, before this kernel), [1] {2^-k / sx, 2^k sx, sx, 0} written here: sx = the power-of-two scale of
// the LayerNorm output that is split against these weights (range rule of split_terms.h; bound from the norm's gamma / beta over n_ln padded channels, C real ones).
// NT = 2, attention streams (bqkv given): the OUTPUT-PROJECTION tiles are stored in two-term form as well (round 6) - fragment `to` of a projection tile keeps its 1 KiB: lane
// (channel l15, dims 4 lg .. 4 lg + 3 of the head group) holds {hi(w0, w1), hi(w2, w3), lo(w0, w1), lo(w2, w3)} of its four scaled weights, the operand of a 32-deep fp16
// MFMA step whose other four k-slots per lane are zero (fused_attn.h proj_accumulate).  Attention streams carry THREE maxima in slot [0] - max |w| over the Q / K / V tiles
// (their scale 2^k), over the V tiles alone, over the projection tiles (their own scale 2^kp) - and a third slot: [1] = {2^-k / sx, 2^k sx, sx, so}, [2] = {2^-kp / so};
// so = the power-of-two scale of the O^T tiles, from |O| <= max |V| <= C max |w_V| B_x + max |b| (a softmax row is a convex combination of V rows).
__global__ __launch_bounds__(256) void attn_x3_pack_kernel(const f32x4* __restrict__ waf, bf16x8* __restrict__ out, int n_tiles, int TPG, int KK, int KS, int TF, unsigned proj_mask, int NT = 3,
                                                           const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr, int n_ln = 0, int C = 0,
                                                           const float* __restrict__ bqkv = nullptr, int n_bqkv = 0) {
    const int per_tile = (KS > KK ? KS : KK) * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n_tiles * per_tile) return;
    const int tile = (int)(idx / per_tile), r = (int)(idx - (long long)tile * per_tile), f = r >> 6, lane = r & 63;
    const f32x4* src = waf + (size_t)tile * KK * 64;
    bf16x8* dst = out + (size_t)tile * TF * 64;
    if ((proj_mask >> (tile % TPG)) & 1) {              // projection tile: fp32 fragments, copied - or (two-term attention stream) split lane by lane
        if (f < KK) {
            const f32x4 w = src[f * 64 + lane];
            if (NT == 2 && bqkv) {
                const float sc = x2_scale(reinterpret_cast<const unsigned*>(out + (size_t)n_tiles * TF * 64)[2]);       // the projection tiles' own scale
                const float v[8] = {w[0], w[1], w[2], w[3], 0.f, 0.f, 0.f, 0.f};
                bf16x8 t[2];
                split_terms<2>(v, t, sc);
                const f32x4 hi = __builtin_bit_cast(f32x4, t[0]), lo = __builtin_bit_cast(f32x4, t[1]);
                reinterpret_cast<f32x4*>(dst)[f * 64 + lane] = f32x4{hi[0], hi[1], lo[0], lo[1]};
            } else {
                reinterpret_cast<f32x4*>(dst)[f * 64 + lane] = w;
            }
        }
        return;
    }
    if (f >= KS) return;
    const int l15 = lane & 15, g = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = 2 * f + (g >> 1), lgs = 2 * (g & 1) + (e >> 2);
        v[e] = kk < KK ? src[kk * 64 + 16 * lgs + l15][e & 3] : 0.f;
    }
    if (NT == 3) {
        bf16x8 t0, t1, t2;
        attn_split3(v, t0, t1, t2);
        dst[(f * 3 + 0) * 64 + lane] = t0; dst[(f * 3 + 1) * 64 + lane] = t1; dst[(f * 3 + 2) * 64 + lane] = t2;
    } else {
        bf16x8* tail = out + (size_t)n_tiles * TF * 64;
        const float sc = x2_scale(*reinterpret_cast<const unsigned*>(tail));
        bf16x8 t[2];
        split_terms<2>(v, t, sc);
        dst[(f * 2 + 0) * 64 + lane] = t[0]; dst[(f * 2 + 1) * 64 + lane] = t[1];
        if (idx == 0) {
            const float bx = gamma ? ln_out_bound(gamma, beta, n_ln, C) : 0.f;
            const float sx = gamma ? act_pow2_scale(bx) : 1.0f;
            float so = 1.0f;
            const unsigned* mxs = reinterpret_cast<const unsigned*>(tail);
            if (bqkv && gamma) {
                float mb = 0.f;
                for (int i = 0; i < n_bqkv; ++i) mb = fmaxf(mb, fabsf(bqkv[i]));
                so = act_pow2_scale((float)C * __uint_as_float(mxs[1]) * bx + mb);
            }
            float* o = reinterpret_cast<float*>(tail + 1); o[0] = (1.0f / sc) * (1.0f / sx); o[1] = sc * sx; o[2] = sx; o[3] = so;
            if (bqkv) { o[4] = (1.0f / x2_scale(mxs[2])) * (1.0f / so); o[5] = 0.f; o[6] = 0.f; o[7] = 0.f; }
        }
    }
}

struct AttnArgs {
    const float* src;           // [B*tokens][CP] block input (also the shortcut)
    float* dst;                 // [B*tokens][CP] output (may alias src: windows own disjoint tokens)
    const float* gamma; const float* beta;
    const f32x4* wf;            // weight stream: [group][tile][KK][64] fragments
    const float* bqkv;          // [group][3*NT][16] biases of the Q/K/V tiles, in stream order
    const float* bias_tab;      // [heads padded][16][16]
    const float* bproj;         // [CP]
    const int* map;             // [slots] window slot -> token or -1
    int slots, tokens, n_windows, nWh, nWw, shifted, C, n_groups;
    float scale, eps;
    // Head-group split (same idea as MlpArgs::HS): workgroup (wb, gs) walks head groups [gs*n_groups/GS, (gs+1)*n_groups/GS) and
    // stores its projection partial sums to partial[gs][rows][CP]; rows_combine_kernel adds them in fixed order.
    int GS; float* partial; int rows;
    unsigned long long* trace;  // tuning builds only (-DESCX_ATTN_TRACE): per-wave cycle sums of the head-group phases
    // Combine-on-load (COMB instantiations, round 4): the block input is still split over the comb_n fc2 partial-sum slabs of the PREVIOUS block's
    // hidden-split MLP.  x = src + (((P0 + P1) + ...) + bias) - the arithmetic of rows_combine_kernel, bit for bit - is formed wherever a row is
    // read (LayerNorm input and shortcut), which removes the combine launch and one write + one read of the map between the two blocks.
    const float* comb_partial; const float* comb_bias; long long comb_stride; int comb_n;
    // TAPE instantiations (training forward, round 4): what the hand-written backward reads is written out as it is produced, in the unfused
    // path's layouts - the normalised rows in window-slot order [slots][CP], q (scaled) | k | v per head in [slots][ldq] with hdp columns per
    // head (q at h*hdp, k at nH*hdp + h*hdp, v at 2*nH*hdp + h*hdp), the attention output [slots][ldo] - so LayerNorm, QKV projection,
    // window attention and output projection are ONE launch and nothing is read back (train.hip, layer_fwd).
    float* tape_xn; float* tape_qkv; float* tape_o; int ldq, ldo, hdp, nH;
    const void* x3_wf;          // X3 instantiations: the split weight stream (attn_x3_pack_kernel), else unused
    int x3_pairs;               // the stream is in pair order [Q0 K0 V0 Q1 K1 V1 P_lo P_hi] (attn_x3p_pack_kernel): X3P instantiations
    const float* x3_scale;      // NT = 2 instantiations: {2^-k / sx, 2^k sx, sx, so | 2^-kp / so} of the block's weight stream (its last 32 bytes; sx / so = scales of the LayerNorm output / of the O^T tiles, split_terms.h)
};

// One 16-byte piece of a block-input row: plain, or combined on the fly from the hidden-split MLP's slabs (see AttnArgs).
template <bool COMB, class Args>
__device__ __forceinline__ f32x4 load_block_input(const Args& a, size_t elem, int col) {
    f32x4 x = ld4(a.src + elem);
    if constexpr (COMB) {
        f32x4 v = ld4(a.comb_partial + elem);
        for (int h = 1; h < a.comb_n; ++h) v += ld4(a.comb_partial + (size_t)h * a.comb_stride + elem);
        v += ld4(a.comb_bias + col);
        x = x + v;
    }
    return x;
}

// Shift mask of one window for this lane's (query l15, keys 4lg..4lg+3): -100 where query and key carry different region labels
// (attention.py:56-75, 233-236).  It only depends on the window, so it is built once per window, not once per head.
__device__ __forceinline__ f32x4 window_shift_mask(int l15, int lg, bool lastH, bool lastW) {
    const int qh = l15 >> 2, qw = l15 & 3;
    const int labq = 3 * (lastH ? (qh < 2 ? 1 : 2) : 0) + (lastW ? (qw < 2 ? 1 : 2) : 0);
    const int labkh = 3 * (lastH ? (lg < 2 ? 1 : 2) : 0);
    f32x4 m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int labk = labkh + (lastW ? (r < 2 ? 1 : 2) : 0);
        m[r] = (labk != labq) ? -100.0f : 0.0f;
    }
    return m;
}

// softmax over the 16 keys of one query row held as 4 values x 4 lane groups (attention.py:226-238)
__device__ __forceinline__ f32x4 window_softmax(f32x4 s, f32x4 bias_row, f32x4 shift_mask, bool shifted) {
    s += bias_row;              // bias_tab[head][query l15][keys 4lg..4lg+3], loaded by the caller a whole GEMM ahead of its use
    if (shifted) s += shift_mask;
    float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    mx = max_groups(mx);
    const float ml = -mx * 1.44269504088896341f;
    f32x4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], 1.44269504088896341f, ml));       // exp(s - max)
    float den = (p[0] + p[1]) + (p[2] + p[3]);
    den = sum_groups(den);
    const float inv = __builtin_amdgcn_rcpf(den);      // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] *= inv;
    return p;
}

// register budget capped for 3-4 waves per SIMD where the tile sizes allow (same reasoning as the fused MLP)
#ifndef ESCX_ATTN_OCC3
#define ESCX_ATTN_OCC3 160
#endif
template <int CP, int TMW> constexpr int attn_min_waves() { return (CP * TMW <= 96) ? 4 : ((CP * TMW <= ESCX_ATTN_OCC3) ? 3 : ((CP * TMW <= 192) ? 2 : 1)); }

// X3 instantiations hold the LayerNorm output as three bf16 terms (1.5x the registers of the fp32 operand): one occupancy step down
template <int CP, int TMW, bool X3> constexpr int attn_min_waves_x() { return !X3 ? attn_min_waves<CP, TMW>() : ((CP * TMW <= 96) ? 3 : ((CP * TMW <= 160) ? 2 : 1)); }

// X3P (with X3, MODE 0 / 1, an even number of head groups per workgroup): the OUTPUT PROJECTION in split form too.  Its contraction is only 16 deep per head group, so two
// consecutive groups form one 32-deep step: the two O^T accumulator tiles of a lane are the 8 k-slots (as the two fc1 tiles are in fused_mlp_x3.h).  Stream per PAIR of groups:
// [Q0 K0 V0 Q1 K1 V1 P_lo P_hi] - the same eight tiles as two groups of [Q K V P]; P_lo / P_hi hold, per output tile of the first / second half, the three split fragments
// -> lane (channel, g): term i of Wp[channel][dims of group 0: 4 g + e | group 1: 4 g + e - 4].
template <int CP, int MODE, int UT, int TMW, int NW, bool COMB = false, bool TAPE = false, bool X3 = false, bool X3P = false, int NT = 3>
__global__ __launch_bounds__(64 * NW, (attn_min_waves_x<CP, TMW, X3>())) void attn_fused_kernel(AttnArgs a) {
    static_assert(!X3 || (!COMB && !TAPE), "the split-operand form exists for the plain inference instantiations");
    static_assert(NT == 3 || (NT == 2 && X3 && !X3P), "two fp16 terms: the split-operand instantiations without the pair projection");
    static_assert(!X3P || (X3 && MODE != 2), "pair projection: split-operand instantiations with one tile per head group");
#ifdef ESCX_ATTN_PRIO
    __builtin_amdgcn_s_setprio(ESCX_ATTN_PRIO);     // tuning builds: static wave priority against co-running launches of the other batch part
#endif
    constexpr int KK = CP / 16;
    constexpr int TPG = (MODE == 2) ? 8 : 4;    // weight tiles per head group
    constexpr int NB = (MODE == 2) ? 6 : 3;     // bias rows per group
    static_assert(TPG % UT == 0, "stage size must divide the tiles of a head group");
    constexpr int KS = attn_x3_ks(CP);
    constexpr int TF = X3 ? attn_x3_tf(CP) : KK;            // fragments (1 KiB) per weight tile in the stream
    __shared__ f32x4 wbuf_static[X3 ? 1 : 2 * UT * KK * 64];
    extern __shared__ __attribute__((aligned(16))) f32x4 wbuf_dyn[];      // X3: 2 * UT * TF KiB (above the 64 KiB of static LDS at the wide layers)
    f32x4* const wbuf0 = X3 ? wbuf_dyn : wbuf_static;

    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: keeps the DMA issue loop scalar
    const int GS = a.GS > 1 ? a.GS : 1;
    const int wgb = blockIdx.x / GS, gs = blockIdx.x - wgb * GS;
    const int g0 = gs * (a.n_groups / GS), g1 = g0 + a.n_groups / GS;
    const int win0 = (wgb * NW + wave) * TMW;
    const int n_stages = (g1 - g0) * (TPG / UT);
    const f32x4* wfb = (X3 ? reinterpret_cast<const f32x4*>(a.x3_wf) : a.wf) + (size_t)g0 * TPG * TF * 64;

    // Weight stream: stage s+1 is DMA'd into the idle LDS buffer while stage s feeds the MFMAs.  The NPW pieces (1 KiB each) a
    // wave owes per stage are not issued in one burst but one per dma_slot() call, which the tile GEMMs make every other
    // k-step.  `slot` is a compile-time constant after unrolling, so the schedule is static and branch-free: the compiler
    // can count the VMEM operations in flight (precise vmcnt for the bias loads) and schedule across the whole stage.
    constexpr int NP = UT * TF, NPW = (NP + NW - 1) / NW;
    const f32x4* dma_src = wfb + lane;
    f32x4* dma_dst = wbuf0;
    int slot = 0;
    auto dma_slot = [&]() {
        if (slot < NPW) {
            int c = wave + slot * NW;
            if (NP % NW != 0) c = min(c, NP - 1);       // tail: a duplicate of the last piece (same bytes, harmless)
            __builtin_amdgcn_global_load_lds((const void*)(dma_src + c * 64), (__attribute__((address_space(3))) void*)(dma_dst + c * 64), 16, 0, 0);
        }
        ++slot;
    };
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma_slot();   // stage 0 up front

    // ---- 1. gather + LayerNorm in registers ------------------------------------------------------
    const int nW = a.nWh * a.nWw;
    f32x4 xf[X3 ? 1 : TMW][X3 ? 1 : KK];
    bf16x8 xs[X3 ? TMW : 1][NT][X3 ? KS : 1];                // X3: LayerNorm output split into NT terms, lane (slot l15, group lg) holds channels 32 s + 8 lg .. + 7
    float x2_dn = 1.f, x2_up = 1.f, x2_sx = 1.f, x2_so = 1.f;   // NT = 2: 2^-k / sx, 2^k sx of the block's scaled weights, sx / so = the power-of-two scales of the LayerNorm output / the O^T tiles
    if constexpr (NT == 2) { x2_dn = a.x3_scale[0]; x2_up = a.x3_scale[1]; x2_sx = a.x3_scale[2]; x2_so = a.x3_scale[3]; }
    int tok[TMW];                               // this lane's token row (element offset / CP), or -1
    bool lastH[TMW], lastW[TMW];
#pragma unroll
    for (int t = 0; t < TMW; ++t) {
        const int win = win0 + t;
        tok[t] = -1; lastH[t] = false; lastW[t] = false;
        if (win < a.n_windows) {
            const int b = win / nW, wloc = win - b * nW;
            const int wh = wloc / a.nWw, ww = wloc - wh * a.nWw;
            lastH[t] = (wh == a.nWh - 1); lastW[t] = (ww == a.nWw - 1);
            const int tk = a.map[wloc * 16 + l15];
            if (tk >= 0) tok[t] = b * a.tokens + tk;
        }
        if constexpr (X3) {
            const float* xrow = a.src + (size_t)(tok[t] < 0 ? 0 : tok[t]) * CP;
            float xv[KS][8];
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = 32 * ks + 8 * lg;
                f32x4 v0 = zero4(), v1 = zero4();
                if (c0 < CP) { v0 = ld4(xrow + c0); v1 = ld4(xrow + c0 + 4); }
                if (tok[t] < 0) { v0 = zero4(); v1 = zero4(); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[ks][e] = v0[e]; xv[ks][4 + e] = v1[e]; s += v0[e]; s += v1[e]; }
            }
            s = sum_groups(s);
            const float mean = s / (float)a.C;
            float v = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xv[ks][e] - mean; v += d * d; }
            v = sum_groups(v) - (float)(32 * KS - a.C) * mean * mean;      // every zero slot (channel padding, K padding to 32) added mean^2
            const float rstd = 1.0f / sqrtf(v / (float)a.C + a.eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = 32 * ks + 8 * lg;
                f32x4 g0v = zero4(), g1v = zero4(), b0v = zero4(), b1v = zero4();
                if (c0 < CP) { g0v = ld4(a.gamma + c0); g1v = ld4(a.gamma + c0 + 4); b0v = ld4(a.beta + c0); b1v = ld4(a.beta + c0 + 4); }
                float xn[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xn[e] = tok[t] >= 0 ? (xv[ks][e] - mean) * rstd * g0v[e] + b0v[e] : 0.f;            // padded slots become zero rows AFTER the norm
                    xn[4 + e] = tok[t] >= 0 ? (xv[ks][4 + e] - mean) * rstd * g1v[e] + b1v[e] : 0.f;
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
        if constexpr (!X3) {
        const size_t xoff = (size_t)(tok[t] < 0 ? 0 : tok[t]) * CP + 4 * lg;
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            xf[t][kk] = tok[t] >= 0 ? load_block_input<COMB>(a, xoff + 16 * kk, 16 * kk + 4 * lg) : zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e) s += xf[t][kk][e];            // pad channels are exact zeros
        }
        s = sum_groups(s);
        const float mean = s / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xf[t][kk][e] - mean; v += d * d; }
        v = sum_groups(v) - (float)(CP - a.C) * mean * mean;     // the zero pads each added mean^2
        const float rstd = 1.0f / sqrtf(v / (float)a.C + a.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 g = ld4(a.gamma + 16 * kk + 4 * lg), bb = ld4(a.beta + 16 * kk + 4 * lg);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xf[t][kk][e] = tok[t] >= 0 ? (xf[t][kk][e] - mean) * rstd * g[e] + bb[e] : 0.f;   // gamma = beta = 0 in the pad channels
        }
        }
    }

    if constexpr (TAPE) {           // the normalised rows, window-slot order (zeros in padded slots), + zeros in the pad columns of the q|k|v and output rows
#pragma unroll
        for (int t = 0; t < TMW; ++t) {
            if (win0 + t >= a.n_windows) continue;
            const size_t row = (size_t)(win0 + t) * 16 + l15;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) st4(a.tape_xn + row * CP + 16 * kk + 4 * lg, xf[t][kk]);
            const int q_end = 3 * a.nH * a.hdp, o_end = a.nH * a.hdp;
            for (int c = q_end + 4 * lg; c < a.ldq; c += 16) st4(a.tape_qkv + row * a.ldq + c, zero4());       // the WHOLE pad width (it used to stop after 16 columns)
            for (int c = o_end + 4 * lg; c < a.ldo; c += 16) st4(a.tape_o + row * a.ldo + c, zero4());
        }
    }

    f32x4 acc[KK][TMW];
#pragma unroll
    for (int o = 0; o < KK; ++o)
#pragma unroll
        for (int t = 0; t < TMW; ++t) acc[o][t] = zero4();

    // Tape stores of one head-group tile (TAPE).  Tile row i of group g (half `half` in MODE 2) is (head, dim) = MODE 0: (g, i); MODE 1: (2g + i / 8, i % 8);
    // MODE 2: (g, 16 half + i) - the weight packer's hd_of - and lives at column head * hdp + dim of its q / k / v / o block (hdp = head_dim rounded up to 4).
    auto tape_col = [&](int g, int half, int i, bool* ok) -> int {
        const int head = (MODE == 1) ? 2 * g + (i >> 3) : g;
        const int dim = (MODE == 1) ? (i & 7) : ((MODE == 2) ? 16 * half + i : i);
        *ok = head < a.nH && dim < a.hdp;
        return head * a.hdp + dim;
    };
    auto tape_rows = [&](float* basep, int ld, int block_col, int g, int half, const f32x4* v) {       // lane (slot l15, tile rows 4lg .. 4lg + 3)
        bool ok; const int c = tape_col(g, half, 4 * lg, &ok);
#pragma unroll
        for (int t = 0; t < TMW; ++t)
            if (win0 + t < a.n_windows && ok) st4(basep + ((size_t)(win0 + t) * 16 + l15) * ld + block_col + c, v[t]);
    };
    auto tape_cols = [&](float* basep, int ld, int block_col, int g, int half, const f32x4* vt) {      // transposed tile: lane (tile row l15, slots 4lg .. 4lg + 3)
        bool ok; const int c = tape_col(g, half, l15, &ok);
#pragma unroll
        for (int t = 0; t < TMW; ++t)
            if (win0 + t < a.n_windows && ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) basep[((size_t)(win0 + t) * 16 + 4 * lg + r) * ld + block_col + c] = vt[t][r];
            }
    };

    // LDS -> register fragment ring over the whole stage (UT tiles = NP consecutive 1 KiB fragments): the read of fragment
    // f + PD is in flight while fragment f feeds the MFMAs, across tile boundaries; only the first PD reads of a stage are
    // exposed.  `frag` and `slot` are compile-time constants after unrolling.  The sched_group_barrier sequence pins the order
    // (hipcc otherwise sinks each ds_read to its first use and waits, and sinks every DMA piece to the end of the stage where
    // the vmcnt(0) before the next barrier would wait out the whole L2 round trip).
    constexpr int PD = X3 ? 1 : (NP < 3 ? NP : 3);         // X3 reads its fragments straight from LDS (no register ring)
    // (the pinned schedule below was tuned for the fp32 instruction stream; the X3 instantiations leave the order to the compiler)
#define ESCX_SGB_DS() do { if constexpr (!X3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } while (0)
#define ESCX_SGB_VMEM(n) do { if constexpr (!X3) __builtin_amdgcn_sched_group_barrier(0x020, n, 0); } while (0)
#define ESCX_SGB_MFMA(n) do { if constexpr (!X3) __builtin_amdgcn_sched_group_barrier(0x008, n, 0); } while (0)
    int stage = 0, frag = 0;
    const f32x4* wb = nullptr;
    f32x4 ring[PD];
    auto next_stage = [&]() {
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma_slot();               // whatever is left of the pending stage (none when the GEMMs offered enough slots)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef ESCX_ATTN_NOBARRIER     // timing experiment only (results are wrong without the barrier)
        __syncthreads();
#endif
        // arm the DMA of the following stage; after the last one its own tiles are re-loaded into the idle buffer (harmless)
        dma_src = wfb + (size_t)min(stage + 1, n_stages - 1) * (UT * TF * 64) + lane;
        dma_dst = wbuf0 + ((stage + 1) & 1) * (NP * 64);
        slot = 0;
        wb = wbuf0 + (stage & 1) * (NP * 64) + lane;
        ++stage;
        frag = 0;
        if constexpr (!X3) {
#pragma unroll
        for (int i = 0; i < PD; ++i) { ring[i] = wb[i * 64]; ESCX_SGB_DS(); }
        }
    };
    auto next_frag = [&]() -> f32x4 {
        const f32x4 w = ring[frag % PD];
        if (frag + PD < NP) { ring[frag % PD] = wb[(frag + PD) * 64]; ESCX_SGB_DS(); }
        ++frag;
        return w;
    };
    auto dma_pinned = [&]() {
        const bool issues = slot < NPW;
        dma_slot();
        if (issues && !X3) ESCX_SGB_VMEM(1);
    };
    int tile = 0;
    auto begin_tile = [&]() { if (tile % UT == 0) next_stage(); ++tile; };
    // X3: the Q / K / V tile GEMMs on the bf16 MFMA, six cross terms (weight term I, activation term J) smallest first, two accumulator chains
#define ESCX_ATTN_X3_TERMS(M) M(0, 2) M(2, 0) M(1, 1) M(0, 1) M(1, 0) M(0, 0)
    auto gemm_x3 = [&](f32x4* out, f32x4 init, bool x_rows) {
        const bf16x8* tb = reinterpret_cast<const bf16x8*>(wb) + (size_t)(((tile - 1) % UT) * TF) * 64;
        f32x4 o2[TMW];
#pragma unroll
        for (int t = 0; t < TMW; ++t) { out[t] = NT == 2 ? init * x2_up : init; o2[t] = zero4(); }       // NT = 2: the sums carry 2^k, so does the bias (exact)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 w[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) w[i] = tb[(ks * NT + i) * 64];
            dma_pinned();
            int n = 0;
#define ESCX_ATTN_X3_STEP(I, J) \
            _Pragma("unroll") for (int t = 0; t < TMW; ++t) { \
                f32x4& d = (n & 1) ? o2[t] : out[t]; \
                d = x_rows ? mma_x<NT>(xs[t][J][ks], w[I], d) : mma_x<NT>(w[I], xs[t][J][ks], d); \
            } ++n;
            if constexpr (NT == 3) { ESCX_ATTN_X3_TERMS(ESCX_ATTN_X3_STEP) } else { ESCX_X2_TERMS(ESCX_ATTN_X3_STEP) }
#undef ESCX_ATTN_X3_STEP
        }
#pragma unroll
        for (int t = 0; t < TMW; ++t) { out[t] += o2[t]; if constexpr (NT == 2) out[t] *= x2_dn; }
    };
    // tile GEMM with the weight tile as the row operand: out[t] = W_tile . x^T  -> lane (token l15, rows 4lg + r)
    auto gemm_w_rows = [&](f32x4* out, f32x4 init) {       // init: the tile's bias rides in the accumulator
        if constexpr (X3) { gemm_x3(out, init, false); return; }
        f32x4 o2[TMW];
#pragma unroll
        for (int t = 0; t < TMW; ++t) { out[t] = init; o2[t] = zero4(); }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 w = next_frag();
            if ((kk & 1) == 0) dma_pinned();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TMW; ++t) {
                    if (TMW == 1 && (r & 1)) o2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], o2[t], 0, 0, 0);
                    else out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], out[t], 0, 0, 0);
                }
            ESCX_SGB_MFMA(4 * TMW);
        }
        if (TMW == 1) out[0] += o2[0];
    };
    // swapped: out[t] = x . W_tile^T -> lane (row l15 of the weight tile, tokens 4lg + r)
    auto gemm_x_rows = [&](f32x4* out, f32x4 init) {
        if constexpr (X3) { gemm_x3(out, init, true); return; }
        f32x4 o2[TMW];
#pragma unroll
        for (int t = 0; t < TMW; ++t) { out[t] = init; o2[t] = zero4(); }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 w = next_frag();
            if ((kk & 1) == 0) dma_pinned();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TMW; ++t) {
                    if (TMW == 1 && (r & 1)) o2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[t][kk][r], w[r], o2[t], 0, 0, 0);
                    else out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[t][kk][r], w[r], out[t], 0, 0, 0);
                }
            ESCX_SGB_MFMA(4 * TMW);
        }
        if (TMW == 1) out[0] += o2[0];
    };
    auto proj_accumulate = [&](const f32x4* o) {
        const f32x4* tb = wb + (size_t)(((tile - 1) % UT) * TF) * 64;       // X3: the projection tile's fp32 fragments, read straight from LDS
        if constexpr (X3 && NT == 2) {
            // Two-term projection (round 6): the head group's 16 dims are the REAL half of a 32-deep fp16 step - a lane's four O^T values (dims 4 lg + r) fill four of its
            // eight k-slots, the other four are zero on both operands, so no value changes lane.  3 MFMAs of 16 cycles per output tile instead of 4 fp32 MFMAs of 32.
            bf16x8 os[TMW][2];
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                const float v8[8] = {o[t][0], o[t][1], o[t][2], o[t][3], 0.f, 0.f, 0.f, 0.f};
                split_terms<2>(v8, os[t], x2_so);
            }
#pragma unroll
            for (int to = 0; to < KK; to += 2) {
                const f32x4 w = tb[to * 64], wn = (to + 1 < KK) ? tb[(to + 1) * 64] : zero4();
                dma_pinned();
                const bf16x8 wh = __builtin_bit_cast(bf16x8, f32x4{w[0], w[1], 0.f, 0.f}), wl = __builtin_bit_cast(bf16x8, f32x4{w[2], w[3], 0.f, 0.f});
                const bf16x8 wnh = __builtin_bit_cast(bf16x8, f32x4{wn[0], wn[1], 0.f, 0.f}), wnl = __builtin_bit_cast(bf16x8, f32x4{wn[2], wn[3], 0.f, 0.f});
#define ESCX_PJ2(WH, WL, TO) _Pragma("unroll") for (int t = 0; t < TMW; ++t) acc[TO][t] = mma_x<2>(WH, os[t][1], acc[TO][t]); \
                             _Pragma("unroll") for (int t = 0; t < TMW; ++t) acc[TO][t] = mma_x<2>(WL, os[t][0], acc[TO][t]); \
                             _Pragma("unroll") for (int t = 0; t < TMW; ++t) acc[TO][t] = mma_x<2>(WH, os[t][0], acc[TO][t]);
                if (to + 1 < KK) {      // two output tiles interleaved: no back-to-back MFMAs on one accumulator
#pragma unroll
                    for (int t = 0; t < TMW; ++t) { acc[to][t] = mma_x<2>(wh, os[t][1], acc[to][t]); acc[to + 1][t] = mma_x<2>(wnh, os[t][1], acc[to + 1][t]); }
#pragma unroll
                    for (int t = 0; t < TMW; ++t) { acc[to][t] = mma_x<2>(wl, os[t][0], acc[to][t]); acc[to + 1][t] = mma_x<2>(wnl, os[t][0], acc[to + 1][t]); }
#pragma unroll
                    for (int t = 0; t < TMW; ++t) { acc[to][t] = mma_x<2>(wh, os[t][0], acc[to][t]); acc[to + 1][t] = mma_x<2>(wnh, os[t][0], acc[to + 1][t]); }
                } else { ESCX_PJ2(wh, wl, to) }
#undef ESCX_PJ2
            }
            return;
        }
#pragma unroll
        for (int to = 0; to < KK; to += 2) {    // two output tiles per step: no back-to-back MFMAs on one accumulator
            f32x4 w, wn = zero4();
            if constexpr (X3) { w = tb[to * 64]; if (to + 1 < KK) wn = tb[(to + 1) * 64]; }
            else { w = next_frag(); if (to + 1 < KK) wn = next_frag(); }
            dma_pinned();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TMW; ++t) {
                    acc[to][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], o[t][r], acc[to][t], 0, 0, 0);
                    if (to + 1 < KK) acc[to + 1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[r], o[t][r], acc[to + 1][t], 0, 0, 0);
                }
            if constexpr (!X3) { if (to + 1 < KK) ESCX_SGB_MFMA(8 * TMW); else ESCX_SGB_MFMA(4 * TMW); }
        }
    };
    // X3P: half of the output tiles of the pair projection (tiles [lo, hi)), six cross terms per K = 32 step, two output tiles at a time
    auto proj_pair = [&](auto lo_c, auto hi_c, const bf16x8 (*hs)[3]) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;         // compile-time: acc[] must stay in registers
        const bf16x8* tb = reinterpret_cast<const bf16x8*>(wb) + (size_t)(((tile - 1) % UT) * TF) * 64;
#pragma unroll
        for (int j = 0; j < hi - lo; j += 2) {
            constexpr int dummy = 0; (void)dummy;
            const int to = lo + j;
            bf16x8 w[3], wn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { w[i] = tb[(j * 3 + i) * 64]; wn[i] = tb[((j + 1 < hi - lo ? j + 1 : j) * 3 + i) * 64]; }
            dma_pinned();
#define ESCX_PP(I, J) _Pragma("unroll") for (int t = 0; t < TMW; ++t) { acc[to][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[I], hs[t][J], acc[to][t], 0, 0, 0); \
                                                     if (to + 1 < hi) acc[to + 1 < hi ? to + 1 : to][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[I], hs[t][J], acc[to + 1 < hi ? to + 1 : to][t], 0, 0, 0); }
            ESCX_PP(0, 2) ESCX_PP(2, 0) ESCX_PP(1, 1) ESCX_PP(0, 1) ESCX_PP(1, 0) ESCX_PP(0, 0)
#undef ESCX_PP
        }
    };

    // Per-group constants (q/k/v biases, relative-position bias rows) are fetched one head group ahead, right behind a stage
    // barrier: they land under a whole group of MFMA work, and the no-op pin behind the next group's first barrier (whose
    // vmcnt(0) has already drained everything) keeps the compiler from waiting at their first use with this stage's DMA in flight.
    struct GroupConst { f32x4 b[4]; f32x4 bt0, bt1; float bv0, bv1; };
    auto load_consts = [&](int g) -> GroupConst {
        GroupConst c;
        const float* bg = a.bqkv + (size_t)g * NB * 16;
        if constexpr (MODE != 2) {
            c.b[0] = ld4(bg + 4 * lg); c.b[1] = ld4(bg + 16 + 4 * lg); c.b[2] = zero4(); c.b[3] = zero4();
            c.bv0 = bg[32 + l15]; c.bv1 = 0.f;
            const float* bt = a.bias_tab + (size_t)(MODE == 1 ? 2 * g : g) * 256 + l15 * 16 + 4 * lg;
            c.bt0 = ld4(bt); c.bt1 = (MODE == 1) ? ld4(bt + 256) : zero4();
            ESCX_SGB_VMEM(MODE == 1 ? 5 : 4);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) c.b[i] = ld4(bg + i * 16 + 4 * lg);
            c.bv0 = bg[4 * 16 + l15]; c.bv1 = bg[5 * 16 + l15];
            c.bt0 = ld4(a.bias_tab + (size_t)g * 256 + l15 * 16 + 4 * lg); c.bt1 = zero4();
            ESCX_SGB_VMEM(7);
        }
        return c;
    };
    auto pin_consts = [&](GroupConst& c) {
        asm volatile("" : "+v"(c.b[0]), "+v"(c.b[1]), "+v"(c.bt0), "+v"(c.bv0));
        if constexpr (MODE == 1) asm volatile("" : "+v"(c.bt1));
        if constexpr (MODE == 2) asm volatile("" : "+v"(c.b[2]), "+v"(c.b[3]), "+v"(c.bv1));
    };

    // ---- 2. head groups ----------------------------------------------------------------------------
#ifdef ESCX_ATTN_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ESCX_TS(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    ESCX_TS(t_begin)
#else
#define ESCX_TS(var)
#endif
    f32x4 shmask[TMW];
#pragma unroll
    for (int t = 0; t < TMW; ++t) shmask[t] = window_shift_mask(l15, lg, lastH[t], lastW[t]);
    GroupConst cur = load_consts(g0);
    f32x4 o_prev[X3P ? TMW : 1];                // X3P: the first group's O^T tile waits for its partner
    for (int g = g0; g < g1; ++g) {
        if (!X3P || ((g - g0) & 1) == 0) tile = 0;
        ESCX_TS(t0)
        begin_tile();                           // stage barrier
        pin_consts(cur);
        const GroupConst nxt = load_consts(min(g + 1, g1 - 1));
        ESCX_TS(t1)
        f32x4 q[TMW], k[TMW], vt[TMW], o[TMW];
        if constexpr (MODE != 2) {
            gemm_w_rows(q, cur.b[0]);
#pragma unroll
            for (int t = 0; t < TMW; ++t) q[t] *= a.scale;
            begin_tile();
            gemm_w_rows(k, cur.b[1]);
            if constexpr (TAPE) {
                tape_rows(a.tape_qkv, a.ldq, 0, g, 0, q);
                tape_rows(a.tape_qkv, a.ldq, a.nH * a.hdp, g, 0, k);
            }
            ESCX_TS(t2)
            f32x4 p0[TMW], p1[TMW];
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                if constexpr (MODE == 0) {
                    f32x4 s = zero4();
#pragma unroll
                    for (int r = 0; r < 4; ++r) s = __builtin_amdgcn_mfma_f32_16x16x4f32(k[t][r], q[t][r], s, 0, 0, 0);
                    p0[t] = window_softmax(s, cur.bt0, shmask[t], a.shifted);
                } else {        // two heads share the tile: head A on k-slot groups 0,1 and head B on 2,3
                    f32x4 sa = zero4(), sb = zero4();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k[t][r], lg < 2 ? q[t][r] : 0.f, sa, 0, 0, 0);
                        sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k[t][r], lg < 2 ? 0.f : q[t][r], sb, 0, 0, 0);
                    }
                    p0[t] = window_softmax(sa, cur.bt0, shmask[t], a.shifted);
                    p1[t] = window_softmax(sb, cur.bt1, shmask[t], a.shifted);
                }
            }
            ESCX_SGB_MFMA((MODE == 0 ? 4 : 8) * TMW);
            ESCX_TS(t3)
            begin_tile();
            gemm_x_rows(vt, f32x4{cur.bv0, cur.bv0, cur.bv0, cur.bv0});
            ESCX_TS(t4)
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                f32x4 oa = zero4();
#pragma unroll
                for (int r = 0; r < 4; ++r) oa = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[t][r], p0[t][r], oa, 0, 0, 0);
                if constexpr (MODE == 1) {
                    f32x4 ob = zero4();
#pragma unroll
                    for (int r = 0; r < 4; ++r) ob = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[t][r], p1[t][r], ob, 0, 0, 0);
                    o[t] = lg < 2 ? oa : ob;        // lane holds dims 4lg + r: dims 0-7 are head A's, 8-15 head B's
                } else {
                    o[t] = oa;
                }
            }
            ESCX_SGB_MFMA((MODE == 0 ? 4 : 8) * TMW);
            if constexpr (TAPE) {
                tape_cols(a.tape_qkv, a.ldq, 2 * a.nH * a.hdp, g, 0, vt);
                tape_rows(a.tape_o, a.ldo, 0, g, 0, o);
            }
            ESCX_TS(t5)
            if constexpr (X3P) {
                if (((g - g0) & 1) == 0) {
#pragma unroll
                    for (int t = 0; t < TMW; ++t) o_prev[t] = o[t];
                } else {
                    bf16x8 hs[TMW][3];
#pragma unroll
                    for (int t = 0; t < TMW; ++t) {
                        float v8[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v8[e] = o_prev[t][e]; v8[4 + e] = o[t][e]; }
                        attn_split3(v8, hs[t][0], hs[t][1], hs[t][2]);
                    }
                    begin_tile();
                    proj_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, (KK + 1) / 2>{}, hs);
                    begin_tile();
                    proj_pair(std::integral_constant<int, (KK + 1) / 2>{}, std::integral_constant<int, KK>{}, hs);
                }
            } else {
            begin_tile();
            proj_accumulate(o);
            }
            ESCX_TS(t6)
#ifdef ESCX_ATTN_TRACE
            tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2; tr[3] += t4 - t3; tr[4] += t5 - t4; tr[5] += t6 - t5;
#endif
        } else {                // MODE 2: stream order [Q_lo, K_lo, Q_hi, K_hi, V_lo, P_lo, V_hi, P_hi]
            f32x4 s[TMW];
#pragma unroll
            for (int t = 0; t < TMW; ++t) s[t] = zero4();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half) begin_tile();
                gemm_w_rows(q, cur.b[2 * half]);
#pragma unroll
                for (int t = 0; t < TMW; ++t) q[t] *= a.scale;
                begin_tile();
                gemm_w_rows(k, cur.b[2 * half + 1]);
                if constexpr (TAPE) {
                    tape_rows(a.tape_qkv, a.ldq, 0, g, half, q);
                    tape_rows(a.tape_qkv, a.ldq, a.nH * a.hdp, g, half, k);
                }
#pragma unroll
                for (int t = 0; t < TMW; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(k[t][r], q[t][r], s[t], 0, 0, 0);
                ESCX_SGB_MFMA(4 * TMW);
            }
            f32x4 p[TMW];
#pragma unroll
            for (int t = 0; t < TMW; ++t) p[t] = window_softmax(s[t], cur.bt0, shmask[t], a.shifted);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                begin_tile();
                { const float bvh = half ? cur.bv1 : cur.bv0; gemm_x_rows(vt, f32x4{bvh, bvh, bvh, bvh}); }
#pragma unroll
                for (int t = 0; t < TMW; ++t) {
                    o[t] = zero4();
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[t][r], p[t][r], o[t], 0, 0, 0);
                }
                ESCX_SGB_MFMA(4 * TMW);
                if constexpr (TAPE) {
                    tape_cols(a.tape_qkv, a.ldq, 2 * a.nH * a.hdp, g, half, vt);
                    tape_rows(a.tape_o, a.ldo, 0, g, half, o);
                }
                begin_tile();
                proj_accumulate(o);
            }
        }
        cur = nxt;
    }
#ifdef ESCX_ATTN_TRACE
    {
        ESCX_TS(t_end)
        if (lane == 0 && a.trace) {
            unsigned long long* o = a.trace + (size_t)(blockIdx.x * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = tr[i];
            o[6] = t_begin; o[7] = t_end;
        }
    }
#endif
#undef ESCX_TS
#undef ESCX_SGB_DS
#undef ESCX_SGB_VMEM
#undef ESCX_SGB_MFMA

    // ---- 3. bias + shortcut, scatter through the map ----------------------------------------------
    if constexpr (X3 && NT == 2) {              // the two-term projection's sums carry 2^k so: back (exact)
        const float dn_proj = a.x3_scale[4];     // 2^-kp / so
#pragma unroll
        for (int o = 0; o < KK; ++o)
#pragma unroll
            for (int t = 0; t < TMW; ++t) acc[o][t] *= dn_proj;
    }
    if (GS > 1) {
#pragma unroll
        for (int t = 0; t < TMW; ++t) {
            if (tok[t] < 0) continue;
            float* pr = a.partial + ((size_t)gs * a.rows + tok[t]) * CP + 4 * lg;
#pragma unroll
            for (int o = 0; o < KK; ++o) st4(pr + 16 * o, acc[o][t]);
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < TMW; ++t) {
        if (tok[t] < 0) continue;
        const size_t soff = (size_t)tok[t] * CP + 4 * lg;
        float* dr = a.dst + (size_t)tok[t] * CP + 4 * lg;
        f32x4 res[KK];          // all loads, then all stores (dst may alias src: interleaved, every store fences the next load)
#pragma unroll
        for (int o = 0; o < KK; ++o) { res[o] = load_block_input<COMB>(a, soff + 16 * o, 16 * o + 4 * lg); acc[o][t] += ld4(a.bproj + 16 * o + 4 * lg); }
#pragma unroll
        for (int o = 0; o < KK; ++o) st4(dr + 16 * o, res[o] + acc[o][t]);
    }
}

// ------------------------------------------------------------------------------------------------
// Packed variant for the H = 2 scale (ESC's bottom of the decoder, C = 384): every 4x4 window there is 8 real tokens
// plus 8 zero-padded ones (attention.py:139-143), so a window-per-tile kernel spends half of its projection MFMAs on
// rows whose q/k/v are just the bias.  Here one 16-row tile carries the real tokens of TWO windows (rows 0-7: window A,
// rows 8-15: window B).  Q/K/V and the output projection run once per pair; for the 16x16 attention of each window the
// key / value operands are rebuilt in registers: real slots come from the packed tile (a half-row swap for K, a
// lane-group swap for V^T), padded slots are the bias.  Requires H == 2, W % 4 == 0, one head per tile (MODE 0).
// ------------------------------------------------------------------------------------------------
template <int CP, int UT, int NW, bool COMB = false, bool X3 = false, bool X3P = false, int NT = 3>
__global__ __launch_bounds__(64 * NW, (attn_min_waves_x<CP, 1, X3>())) void attn_packed_kernel(AttnArgs a) {
    static_assert(!X3 || !COMB, "the split-operand form exists for the plain instantiation");
    static_assert(NT == 3 || (NT == 2 && X3 && !X3P), "two fp16 terms: the split-operand instantiation without the pair projection");
    static_assert(!X3P || X3, "pair projection: split-operand instantiation");
#ifdef ESCX_ATTN_PRIO
    __builtin_amdgcn_s_setprio(ESCX_ATTN_PRIO);
#endif
    constexpr int KK = CP / 16;
    constexpr int TPG = 4;
    static_assert(TPG % UT == 0, "stage size must divide the tiles of a head group");
    constexpr int KS = attn_x3_ks(CP);
    constexpr int TF = X3 ? attn_x3_tf(CP) : KK;            // fragments per weight tile in the stream (attn_fused_kernel)
    __shared__ f32x4 wbuf_static[X3 ? 1 : 2 * UT * KK * 64];
    extern __shared__ __attribute__((aligned(16))) f32x4 wbuf_dyn[];
    f32x4* const wbuf0 = X3 ? wbuf_dyn : wbuf_static;

    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int GS = a.GS > 1 ? a.GS : 1;
    const int wgb = blockIdx.x / GS, gs = blockIdx.x - wgb * GS;
    const int g0 = gs * (a.n_groups / GS), g1 = g0 + a.n_groups / GS;
    const int pair = wgb * NW + wave;
    const int n_stages = (g1 - g0) * (TPG / UT);
    const f32x4* wfb = (X3 ? reinterpret_cast<const f32x4*>(a.x3_wf) : a.wf) + (size_t)g0 * TPG * TF * 64;

    constexpr int NP = UT * TF, NPW = (NP + NW - 1) / NW;      // static, branch-free DMA schedule as in attn_fused_kernel
    const f32x4* dma_src = wfb + lane;
    f32x4* dma_dst = wbuf0;
    int slot = 0;
    auto dma_slot = [&]() {
        if (slot < NPW) {
            int c = wave + slot * NW;
            if (NP % NW != 0) c = min(c, NP - 1);
            __builtin_amdgcn_global_load_lds((const void*)(dma_src + c * 64), (__attribute__((address_space(3))) void*)(dma_dst + c * 64), 16, 0, 0);
        }
        ++slot;
    };
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma_slot();

    // ---- gather: packed row j <- window (2*pair + j/8), slot (j%8) + off ; off = 8 in shifted blocks (the real rows roll down)
    const int nW = a.nWh * a.nWw;                           // nWh == 1 here
    const int off = a.shifted ? 8 : 0;
    const bool isB = l15 >= 8;
    const int win = 2 * pair + (isB ? 1 : 0);
    const int qslot = (l15 & 7) + off;
    int tok = -1; bool lastW = false;
    if (win < a.n_windows) {
        const int b = win / nW, wloc = win - b * nW;
        lastW = (wloc % a.nWw) == a.nWw - 1;
        const int tk = a.map[wloc * 16 + qslot];
        if (tk >= 0) tok = b * a.tokens + tk;
    }
    f32x4 xf[X3 ? 1 : KK];
    bf16x8 xs[NT][X3 ? KS : 1];
    float x2_dn = 1.f, x2_up = 1.f, x2_sx = 1.f, x2_so = 1.f;
    if constexpr (NT == 2) { x2_dn = a.x3_scale[0]; x2_up = a.x3_scale[1]; x2_sx = a.x3_scale[2]; x2_so = a.x3_scale[3]; }
    if constexpr (X3) {
        const float* xrow = a.src + (size_t)(tok < 0 ? 0 : tok) * CP;
        float xv[KS][8];
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = 32 * ks + 8 * lg;
            f32x4 v0 = zero4(), v1 = zero4();
            if (c0 < CP) { v0 = ld4(xrow + c0); v1 = ld4(xrow + c0 + 4); }
            if (tok < 0) { v0 = zero4(); v1 = zero4(); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[ks][e] = v0[e]; xv[ks][4 + e] = v1[e]; s += v0[e]; s += v1[e]; }
        }
        s = sum_groups(s);
        const float mean = s / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xv[ks][e] - mean; v += d * d; }
        v = sum_groups(v) - (float)(32 * KS - a.C) * mean * mean;
        const float rstd = 1.0f / sqrtf(v / (float)a.C + a.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = 32 * ks + 8 * lg;
            f32x4 g0v = zero4(), g1v = zero4(), b0v = zero4(), b1v = zero4();
            if (c0 < CP) { g0v = ld4(a.gamma + c0); g1v = ld4(a.gamma + c0 + 4); b0v = ld4(a.beta + c0); b1v = ld4(a.beta + c0 + 4); }
            float xn[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xn[e] = tok >= 0 ? (xv[ks][e] - mean) * rstd * g0v[e] + b0v[e] : 0.f;
                xn[4 + e] = tok >= 0 ? (xv[ks][4 + e] - mean) * rstd * g1v[e] + b1v[e] : 0.f;
            }
            if constexpr (NT == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xn[e] *= x2_sx;
            }
            bf16x8 tt[NT];
            split_terms<NT>(xn, tt);
#pragma unroll
            for (int i = 0; i < NT; ++i) xs[i][ks] = tt[i];
        }
    } else {
        const size_t xoff = (size_t)(tok < 0 ? 0 : tok) * CP + 4 * lg;
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            xf[kk] = tok >= 0 ? load_block_input<COMB>(a, xoff + 16 * kk, 16 * kk + 4 * lg) : zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e) s += xf[kk][e];
        }
        s = sum_groups(s);
        const float mean = s / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xf[kk][e] - mean; v += d * d; }
        v = sum_groups(v) - (float)(CP - a.C) * mean * mean;     // the zero pads each added mean^2
        const float rstd = 1.0f / sqrtf(v / (float)a.C + a.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 g = ld4(a.gamma + 16 * kk + 4 * lg), bb = ld4(a.beta + 16 * kk + 4 * lg);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xf[kk][e] = tok >= 0 ? (xf[kk][e] - mean) * rstd * g[e] + bb[e] : 0.f;
        }
    }
    // which lanes hold real keys / values of a window, and whether the packed rows must be moved to reach their slots
    const bool key_real = (l15 >> 3) == (off >> 3);         // K operand: lane = key slot
    const bool val_real = (lg >> 1) == (off >> 3);          // V^T operand: lane group = 4 key slots
    const bool moveA = off != 0, moveB = off != 8;

    f32x4 acc[KK];
#pragma unroll
    for (int o = 0; o < KK; ++o) acc[o] = zero4();

    constexpr int PD = X3 ? 1 : (NP < 3 ? NP : 3);         // stage-long fragment ring + pinned schedule, as in attn_fused_kernel (X3: fragments straight from LDS)
#define ESCX_SGB_DS() do { if constexpr (!X3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } while (0)
#define ESCX_SGB_VMEM(n) do { if constexpr (!X3) __builtin_amdgcn_sched_group_barrier(0x020, n, 0); } while (0)
#define ESCX_SGB_MFMA(n) do { if constexpr (!X3) __builtin_amdgcn_sched_group_barrier(0x008, n, 0); } while (0)
    int stage = 0, frag = 0;
    const f32x4* wb = nullptr;
    f32x4 ring[PD];
    auto next_stage = [&]() {
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma_slot();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_src = wfb + (size_t)min(stage + 1, n_stages - 1) * (UT * TF * 64) + lane;
        dma_dst = wbuf0 + ((stage + 1) & 1) * (NP * 64);
        slot = 0;
        wb = wbuf0 + (stage & 1) * (NP * 64) + lane;
        ++stage;
        frag = 0;
        if constexpr (!X3) {
#pragma unroll
        for (int i = 0; i < PD; ++i) { ring[i] = wb[i * 64]; ESCX_SGB_DS(); }
        }
    };
    auto next_frag = [&]() -> f32x4 {
        const f32x4 w = ring[frag % PD];
        if (frag + PD < NP) { ring[frag % PD] = wb[(frag + PD) * 64]; ESCX_SGB_DS(); }
        ++frag;
        return w;
    };
    auto dma_pinned = [&]() {
        const bool issues = slot < NPW;
        dma_slot();
        if (issues) ESCX_SGB_VMEM(1);
    };
    int tile = 0;
    auto begin_tile = [&]() { if (tile % UT == 0) next_stage(); ++tile; };
    auto tile_gemm = [&](bool x_rows, f32x4 init) -> f32x4 {       // init: the tile's bias rides in the accumulator
        f32x4 o1 = init, o2 = zero4();
        if constexpr (X3) {          // Q / K / V on the bf16 / fp16 MFMA: the cross terms of the split (split_terms.h), smallest first, two accumulator chains
            const bf16x8* tb = reinterpret_cast<const bf16x8*>(wb) + (size_t)(((tile - 1) % UT) * TF) * 64;
            if constexpr (NT == 2) o1 *= x2_up;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 w[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) w[i] = tb[(ks * NT + i) * 64];
                dma_pinned();
#define ESCX_PK_X3(I, J, D) D = x_rows ? mma_x<NT>(xs[J][ks], w[I], D) : mma_x<NT>(w[I], xs[J][ks], D);
                if constexpr (NT == 3) { ESCX_PK_X3(0, 2, o1) ESCX_PK_X3(2, 0, o2) ESCX_PK_X3(1, 1, o1) ESCX_PK_X3(0, 1, o2) ESCX_PK_X3(1, 0, o1) ESCX_PK_X3(0, 0, o2) }
                else { ESCX_PK_X3(0, 1, o1) ESCX_PK_X3(1, 0, o2) ESCX_PK_X3(0, 0, o1) }
#undef ESCX_PK_X3
            }
            if constexpr (NT == 2) return (o1 + o2) * x2_dn;
            return o1 + o2;
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 w = next_frag();
            if ((kk & 1) == 0) dma_pinned();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (x_rows) {
                    if (r & 1) o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[kk][r], w[r], o2, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[kk][r], w[r], o1, 0, 0, 0);
                } else {
                    if (r & 1) o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[kk][r], o2, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[kk][r], o1, 0, 0, 0);
                }
            }
            ESCX_SGB_MFMA(4);
        }
        return o1 + o2;
    };
    struct GroupConst { f32x4 bq, bk, bt; float bv; };
    auto load_consts = [&](int g) -> GroupConst {
        const float* bg = a.bqkv + (size_t)g * 3 * 16;
        GroupConst c;
        c.bq = ld4(bg + 4 * lg); c.bk = ld4(bg + 16 + 4 * lg); c.bv = bg[32 + l15];
        c.bt = ld4(a.bias_tab + (size_t)g * 256 + qslot * 16 + 4 * lg);
        ESCX_SGB_VMEM(4);
        return c;
    };

    const f32x4 shmask = window_shift_mask(qslot, lg, true, lastW);
    GroupConst cur = load_consts(g0);
    f32x4 o_prev = zero4();                     // X3P: the first group's O^T tile waits for its partner (attn_fused_kernel)
    for (int g = g0; g < g1; ++g) {
        if (!X3P || ((g - g0) & 1) == 0) tile = 0;
        begin_tile();
        asm volatile("" : "+v"(cur.bq), "+v"(cur.bk), "+v"(cur.bt), "+v"(cur.bv));     // see attn_fused_kernel
        const GroupConst nxt = load_consts(min(g + 1, g1 - 1));
        const f32x4 bk = cur.bk;
        const float bv = cur.bv;
        f32x4 q = tile_gemm(false, cur.bq) * a.scale;
        begin_tile();
        f32x4 k = tile_gemm(false, bk);
        // scores of both windows against every packed query; a lane keeps its own window's row
        f32x4 kA, kB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float km = lane_xor8(k[r]);
            kA[r] = key_real ? (moveA ? km : k[r]) : bk[r];
            kB[r] = key_real ? (moveB ? km : k[r]) : bk[r];
        }
        f32x4 sA = zero4(), sB = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sA = __builtin_amdgcn_mfma_f32_16x16x4f32(kA[r], q[r], sA, 0, 0, 0);
            sB = __builtin_amdgcn_mfma_f32_16x16x4f32(kB[r], q[r], sB, 0, 0, 0);
        }
        ESCX_SGB_MFMA(8);
        const f32x4 p = window_softmax(isB ? sB : sA, cur.bt, shmask, a.shifted);
        begin_tile();
        f32x4 vt = tile_gemm(true, f32x4{bv, bv, bv, bv});
        f32x4 vA, vB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vm = lane_xor32(vt[r], lg >= 2);
            vA[r] = val_real ? (moveA ? vm : vt[r]) : bv;
            vB[r] = val_real ? (moveB ? vm : vt[r]) : bv;
        }
        f32x4 oA = zero4(), oB = zero4();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            oA = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[r], p[r], oA, 0, 0, 0);
            oB = __builtin_amdgcn_mfma_f32_16x16x4f32(vB[r], p[r], oB, 0, 0, 0);
        }
        ESCX_SGB_MFMA(8);
        const f32x4 o = isB ? oB : oA;
        if constexpr (X3P) {
            if (((g - g0) & 1) == 0) { o_prev = o; cur = nxt; continue; }
            float v8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = o_prev[e]; v8[4 + e] = o[e]; }
            bf16x8 hs[3];
            attn_split3(v8, hs[0], hs[1], hs[2]);
            constexpr int H = (KK + 1) / 2;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                begin_tile();
                const bf16x8* tb = reinterpret_cast<const bf16x8*>(wb) + (size_t)(((tile - 1) % UT) * TF) * 64;
                const int lo = half ? H : 0, n = half ? KK - H : H;
#pragma unroll
                for (int j = 0; j < H; j += 2) {
                    if (j < n) {
                        const int to = lo + j;
                        bf16x8 w[3], wn[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) { w[i] = tb[(j * 3 + i) * 64]; wn[i] = tb[((j + 1 < n ? j + 1 : j) * 3 + i) * 64]; }
                        dma_pinned();
#define ESCX_PKP(I, J) acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[I], hs[J], acc[to], 0, 0, 0); \
                       if (j + 1 < n) acc[j + 1 < n ? to + 1 : to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[I], hs[J], acc[j + 1 < n ? to + 1 : to], 0, 0, 0);
                        ESCX_PKP(0, 2) ESCX_PKP(2, 0) ESCX_PKP(1, 1) ESCX_PKP(0, 1) ESCX_PKP(1, 0) ESCX_PKP(0, 0)
#undef ESCX_PKP
                    }
                }
            }
            cur = nxt;
            continue;
        }
        begin_tile();
        const f32x4* ptb = wb + (size_t)(((tile - 1) % UT) * TF) * 64;      // X3: the projection tile's fp32 fragments, read straight from LDS
        if constexpr (X3 && NT == 2) {          // two-term projection on the fp16 MFMA (attn_fused_kernel proj_accumulate)
            const float v8[8] = {o[0], o[1], o[2], o[3], 0.f, 0.f, 0.f, 0.f};
            bf16x8 os[2];
            split_terms<2>(v8, os, x2_so);
#pragma unroll
            for (int to = 0; to < KK; to += 2) {
                const f32x4 w = ptb[to * 64], wn = (to + 1 < KK) ? ptb[(to + 1) * 64] : zero4();
                dma_pinned();
                const bf16x8 wh = __builtin_bit_cast(bf16x8, f32x4{w[0], w[1], 0.f, 0.f}), wl = __builtin_bit_cast(bf16x8, f32x4{w[2], w[3], 0.f, 0.f});
                const bf16x8 wnh = __builtin_bit_cast(bf16x8, f32x4{wn[0], wn[1], 0.f, 0.f}), wnl = __builtin_bit_cast(bf16x8, f32x4{wn[2], wn[3], 0.f, 0.f});
                acc[to] = mma_x<2>(wh, os[1], acc[to]); if (to + 1 < KK) acc[to + 1] = mma_x<2>(wnh, os[1], acc[to + 1]);
                acc[to] = mma_x<2>(wl, os[0], acc[to]); if (to + 1 < KK) acc[to + 1] = mma_x<2>(wnl, os[0], acc[to + 1]);
                acc[to] = mma_x<2>(wh, os[0], acc[to]); if (to + 1 < KK) acc[to + 1] = mma_x<2>(wnh, os[0], acc[to + 1]);
            }
            cur = nxt;
            continue;
        }
#pragma unroll
        for (int to = 0; to < KK; to += 2) {
            f32x4 w, wn = zero4();
            if constexpr (X3) { w = ptb[to * 64]; if (to + 1 < KK) wn = ptb[(to + 1) * 64]; }
            else { w = next_frag(); if (to + 1 < KK) wn = next_frag(); }
            dma_pinned();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], o[r], acc[to], 0, 0, 0);
                if (to + 1 < KK) acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[r], o[r], acc[to + 1], 0, 0, 0);
            }
            if (to + 1 < KK) ESCX_SGB_MFMA(8); else ESCX_SGB_MFMA(4);
        }
        cur = nxt;
    }
#undef ESCX_SGB_DS
#undef ESCX_SGB_VMEM
#undef ESCX_SGB_MFMA

    if constexpr (X3 && NT == 2) {
        const float dn_proj = a.x3_scale[4];     // 2^-kp / so
#pragma unroll
        for (int o = 0; o < KK; ++o) acc[o] *= dn_proj;
    }
    if (GS > 1) {
        if (tok >= 0) {
            float* pr = a.partial + ((size_t)gs * a.rows + tok) * CP + 4 * lg;
#pragma unroll
            for (int o = 0; o < KK; ++o) st4(pr + 16 * o, acc[o]);
        }
        return;
    }
    if (tok >= 0) {
        const size_t soff = (size_t)tok * CP + 4 * lg;
        float* dr = a.dst + (size_t)tok * CP + 4 * lg;
        f32x4 res[KK];
#pragma unroll
        for (int o = 0; o < KK; ++o) { res[o] = load_block_input<COMB>(a, soff + 16 * o, 16 * o + 4 * lg); acc[o] += ld4(a.bproj + 16 * o + 4 * lg); }
#pragma unroll
        for (int o = 0; o < KK; ++o) st4(dr + 16 * o, res[o] + acc[o]);
    }
}

}  // namespace escx
