// Fused Swin MLP with fp32 operands SPLIT into three bf16 terms on the bf16 matrix cores (round 5):
//
//   x <- x + W2 . gelu(W1 . LN(x) + b1) + b2,      every fp32 operand a = a1 + a2 + a3 with a_i bf16 (3 x 8 = 24 significand bits: the split is exact)
//
// The product of two bf16 terms is exact in fp32, so  a . b = sum_ij a_i b_j  holds term by term; the six terms a1b1, a1b2, a2b1, a2b2, a1b3, a3b1
// carry everything down to 2^-24 of the product, and they are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (32 products per instruction, ~17 cycles)
// where v_mfma_f32_16x16x4_f32 spends 8 x 32 cycles on the same K.  Measured against fp64 the six-term sum is 50x closer than an fp32 sgemm
// (truncation 6e-9 of the mean magnitude against 3e-7 of rounding): this is fp32-grade arithmetic, not a bf16 approximation -
// the error that remains is the fp32 accumulation, as in any fp32 kernel.  It is NOT bit-identical to the fp32-MFMA kernel (other summation order),
// so the parity evidence is the oracle sweeps, as for every other re-association (tests/test_gpu_parity.py sweeps, clustered codebooks).
//
// Structure = fused_mlp.h: a wave owns 16 token rows end to end in registers; hidden units are walked 32 at a time (two fc1 accumulator tiles ARE
// the 8-deep k-slots of one fc2 step); the split weight fragments ([pair][fragment][lane][8 bf16], built on the device by mlp_x3_pack_kernel from the
// fp32 weights) stream through a double-buffered LDS ring shared by the NW waves; GELU and the operand split run on the VALU beside the matrix cores.
#pragma once
#include <hip/hip_runtime.h>
#include "fused_mlp.h"
#include "gemm_bf16.h"
#include "split_terms.h"

namespace escx {

// a = s[0] + s[1] + s[2] exactly (round-to-nearest-even at each step; the residuals are exact fp32 subtractions)
template <int N>
__device__ __forceinline__ void split3_bf16(const float (&v)[N], __bf16 (&s0)[N], __bf16 (&s1)[N], __bf16 (&s2)[N]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const __bf16 a1 = (__bf16)v[e];
        const float r1 = v[e] - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;
        s0[e] = a1; s1[e] = a2; s2[e] = (__bf16)r2;
    }
}
__device__ __forceinline__ bf16x8 pack8(const __bf16 (&s)[8]) { bf16x8 r; for (int e = 0; e < 8; ++e) r[e] = s[e]; return r; }

// NT = 2: two fp16 terms per operand and three cross products (split_terms.h).
template <int N>
__device__ __forceinline__ void split2_f16(const float (&v)[N], float scale, _Float16 (&s0)[N], _Float16 (&s1)[N]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const float a = v[e] * scale;
        const _Float16 a1 = (_Float16)a;
        s0[e] = a1; s1[e] = (_Float16)(a - (float)a1);
    }
}
__device__ __forceinline__ bf16x8 pack8h(const _Float16 (&s)[8]) { half8 r; for (int e = 0; e < 8; ++e) r[e] = s[e]; return __builtin_bit_cast(bf16x8, r); }
// rows of activations -> NT term fragments
template <int NT>
__device__ __forceinline__ void split_rows(const float (&v)[8], bf16x8 (&out)[NT]) { split_terms<NT>(v, out); }

constexpr int mlp_x3_ks(int CP) { return (CP + 31) / 32; }
constexpr int mlp_x3_frags(int CP, int NT = 3) { return 2 * NT * mlp_x3_ks(CP) + NT * (CP / 16); }       // 1 KiB fragments per pair of hidden tiles: fc1 (KS steps x 2 tiles x NT terms), fc2 (KK tiles x NT terms)
// LDS staging.  Narrow layers (<= 36 fragments per pair): ONE stage per pair.  Wide layers: fc1 in stages of up to 6 K-steps (36 KiB), fc2 in stages of up to 6 pairs of
// output tiles (36 KiB); two ring slots of the largest stage.
constexpr int MLP_X3_STAGE = 36;
constexpr bool mlp_x3_single(int CP, int NT = 3) { return mlp_x3_frags(CP, NT) <= MLP_X3_STAGE; }
constexpr int mlp_x3_gmax(int NT) { return MLP_X3_STAGE / (2 * NT); }               // K-steps (fc1) or output-tile pairs (fc2) per 36 KiB stage: 6 / 9
constexpr int mlp_x3_g1(int CP, int NT = 3) { return mlp_x3_single(CP, NT) ? mlp_x3_ks(CP) : (mlp_x3_ks(CP) <= mlp_x3_gmax(NT) ? mlp_x3_ks(CP) : mlp_x3_gmax(NT)); }                    // K-steps per fc1 stage
constexpr int mlp_x3_g2(int CP, int NT = 3) { return mlp_x3_single(CP, NT) ? (CP / 16 + 1) / 2 : ((CP / 16 + 1) / 2 <= mlp_x3_gmax(NT) ? (CP / 16 + 1) / 2 : mlp_x3_gmax(NT)); }           // output-tile pairs per fc2 stage
constexpr int mlp_x3_stage_frags(int CP, int NT = 3) {
    return mlp_x3_single(CP, NT) ? mlp_x3_frags(CP, NT) : 2 * NT * (mlp_x3_g1(CP, NT) > mlp_x3_g2(CP, NT) ? mlp_x3_g1(CP, NT) : mlp_x3_g2(CP, NT));
}

// Split image of one block's MLP weights, fragments in CONSUMPTION order.  Pair p of hidden tiles (2p, 2p + 1):
//   fc1 fragment f = (s * 2 + tt) * 3 + i          -> lane (n, g) holds term i of W1[16 (2p + tt) + n][32 s + 8 g + e], e = 0..7 (zero beyond Cp)
//   fc2 fragment f = 6 KS + o * 3 + i              -> lane (c, g) holds term i of W2[16 o + c][unit(g, e)], unit = 16 (2p) + 4 g + e for e < 4, 16 (2p + 1) + 4 g + e - 4 otherwise
// (the k-slot <-> hidden unit map the kernel's two fc1 accumulator tiles dictate).  One thread per (pair, fragment triple, lane).
// NT = 2: the image ends with three 16-byte slots - [0] bits of max |w1|, max |w2|, max_row ||w1_row||^2 (absmax_bits_kernel / rownorm2_max_bits_kernel, before this kernel),
// [1] {2^-k1 / sx, 2^-k2 / sh, 2^k1 sx, 2^k2 sh} and [2] {sx, sh} written here: sx, sh = the power-of-two scales of the LayerNorm output and of the GELU output (split_terms.h
// range rule; bounds from ln2's gamma / beta, the fc1 row norms and b1).
__global__ __launch_bounds__(256) void mlp_x3_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, bf16x8* __restrict__ out, int Cp, int HP, int KS, int KK, int NT,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ b1, int C) {
    const int CH = 2 * NT * KS + NT * KK, triples = 2 * KS + KK;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)(HP / 32) * triples * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63); const long long q = idx >> 6;
    const int t3 = (int)(q % triples), p = (int)(q / triples);
    const int n = lane & 15, g = lane >> 4;
    float v[8];
    int fbase;
    if (t3 < 2 * KS) {
        const int s = t3 >> 1, tt = t3 & 1;
        fbase = t3 * NT;
        const float* row = w1 + (size_t)(16 * (2 * p + tt) + n) * Cp;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = 32 * s + 8 * g + e; v[e] = k < Cp ? row[k] : 0.f; }
    } else {
        const int o = t3 - 2 * KS;
        fbase = 2 * NT * KS + o * NT;
        const float* row = w2 + (size_t)(16 * o + n) * HP;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = row[16 * (2 * p + (e >> 2)) + 4 * g + (e & 3)];
    }
    bf16x8* dst = out + ((size_t)p * CH + fbase) * 64 + lane;
    if (NT == 3) {
        __bf16 s0[8], s1[8], s2[8];
        split3_bf16(v, s0, s1, s2);
        dst[0] = pack8(s0); dst[64] = pack8(s1); dst[128] = pack8(s2);
    } else {
        bf16x8* tail = out + (size_t)(HP / 32) * CH * 64;
        const unsigned* mx = reinterpret_cast<const unsigned*>(tail);
        const float sc1 = x2_scale(mx[0]), sc2 = x2_scale(mx[1]);
        _Float16 h0[8], h1[8];
        split2_f16(v, t3 < 2 * KS ? sc1 : sc2, h0, h1);
        dst[0] = pack8h(h0); dst[64] = pack8h(h1);
        if (idx == 0) {
            float sx = 1.f, sh = 1.f;             // gamma == nullptr: no activation scales (the round-5 form; tagged builds, ESCX_X2_NO_ACT_SCALE)
            if (gamma) {
                float gb = 0.f, mb1 = 0.f;
                const float bx = ln_out_bound(gamma, beta, Cp, C, &gb);
                for (int i = 0; i < HP; ++i) mb1 = fmaxf(mb1, fabsf(b1[i]));
                const float bh = sqrtf(__uint_as_float(mx[2])) * sqrtf((float)C) * gb + mb1;       // |gelu(h)| <= |h| <= ||w1_row|| ||xn|| + |b1|
                sx = act_pow2_scale(bx); sh = act_pow2_scale(bh);
            }
            float* f = reinterpret_cast<float*>(tail + 1);
            f[0] = (1.0f / sc1) * (1.0f / sx); f[1] = (1.0f / sc2) * (1.0f / sh); f[2] = sc1 * sx; f[3] = sc2 * sh;
            f[4] = sx; f[5] = sh; f[6] = 0.f; f[7] = 0.f;
        }
    }
}

// PatchSplit weights for the SPLIT epilogue of mlp_x3_kernel, from the fp32 fragment stream (Layer::sub_wf, [NT][KK][64] float4): fragment (nt, s, i) -> lane (n, g) holds
// term i of W[16 nt + n][16 (2 s + (e >> 2)) + 4 g + (e & 3)], e = 0..7 - the k-slot <-> channel map of the epilogue (two accumulator tiles = one K step).
__global__ __launch_bounds__(256) void mlp_x3_split_pack_kernel(const f32x4* __restrict__ wf, bf16x8* __restrict__ out, int NT, int KK) {
    const int KSY = (KK + 1) / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)NT * KSY * 64) return;
    const int lane = (int)(idx & 63), q = (int)(idx >> 6), s = q % KSY, nt = q / KSY;
    const int l15 = lane & 15, g = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int kk = 2 * s + (e >> 2); v[e] = kk < KK ? wf[((size_t)nt * KK + kk) * 64 + 16 * g + l15][e & 3] : 0.f; }
    __bf16 s0[8], s1[8], s2[8];
    split3_bf16(v, s0, s1, s2);
    bf16x8* dst = out + ((size_t)(nt * KSY + s) * 3) * 64 + lane;
    dst[0] = pack8(s0); dst[64] = pack8(s1); dst[128] = pack8(s2);
}

// Timing-only ablations for tagged builds (-DESCX_X3_ABL=bits; results are wrong by construction, profiles/r5_mlp_x3_ablation.txt): 1 = one cross term instead of six,
// 2 = no GELU and no operand split of the hidden tile (one conversion), 4 = no wait / barrier at the weight stages.
#ifndef ESCX_X3_ABL
#define ESCX_X3_ABL 0
#endif
// the six cross terms (weight term i, activation term j), smallest first
#if ESCX_X3_ABL & 1
#define ESCX_X3_TERMS(M) M(0, 0)
#else
#define ESCX_X3_TERMS(M) M(0, 2) M(2, 0) M(1, 1) M(0, 1) M(1, 0) M(0, 0)
#endif


template <int CP> constexpr int mlp_x3_min_waves() { return CP <= 96 ? 4 : (CP <= 192 ? 2 : 1); }

// SPLIT: PatchSplit in the epilogue (as fused_mlp.h SPLIT): LayerNorm(C) of x + mlp(x) in registers, Linear(C -> 2 C') with split operands, two-row scatter; x is not written.
// Round 6, measured and rejected (profiles/r6_mlp_pipe_ab.txt): a software-pipelined main loop for the one-stage widths - iteration p issuing the fc1 MFMAs of pair p + 1,
// the GELU + split of pair p and the fc2 MFMAs of pair p - 1 as three independent streams (two double-buffered rings, biases loaded one iteration ahead), with a packed
// v_pk_fma_f32 GELU.  Bit-identical, 110 registers, still four waves per SIMD, and 6-18 % SLOWER alone (C = 45 0.884 -> 0.937 ms per step, C = 72 0.468 -> 0.510, C = 96
// 0.473 -> 0.558): on this chip VALU and MFMA issue do not overlap (profiles/r1_ubench_mfma_valu.txt: time = 32 N_mfma + ~4 N_valu at any occupancy; v_pk_fma_f32 costs two
// v_fma_f32), so re-ordering independent work buys nothing and the fill / drain iterations and two more stage barriers per row tile are pure cost.
template <int CP, int NW, bool SPLIT = false, int NT = 3>
__global__ __launch_bounds__(64 * NW, (mlp_x3_min_waves<CP>())) void mlp_x3_kernel(MlpArgs a) {
    constexpr int KS = mlp_x3_ks(CP), KK = CP / 16, CH = mlp_x3_frags(CP, NT), NOP = (KK + 1) / 2;
    constexpr bool SINGLE = mlp_x3_single(CP, NT);
    constexpr int G1 = mlp_x3_g1(CP, NT), G2 = mlp_x3_g2(CP, NT), NS1 = (KS + G1 - 1) / G1, NS2 = (NOP + G2 - 1) / G2;
    constexpr int NSP = SINGLE ? 1 : NS1 + NS2;                                   // stages per pair
    constexpr int SF = mlp_x3_stage_frags(CP, NT);                               // ring slot size in fragments
    constexpr int N2 = 2 * NT;
    extern __shared__ __attribute__((aligned(16))) bf16x8 x3_wbuf[];             // [2][SF * 64]
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // hidden split (fused_mlp.h): workgroup (rb, hs) walks a third of the hidden pairs and writes raw fc2 partial sums; same XCD-aware block order
    const int HS = a.HS > 1 ? a.HS : 1;
    int rb = blockIdx.x, hs = 0;
    if (HS > 1) {
        const int nrb = (a.M + 16 * NW - 1) / (16 * NW), full = nrb & ~7, i = blockIdx.x;
        if (i < full * HS) { const int grp = i / (8 * HS), r = i - grp * (8 * HS); rb = grp * 8 + (r & 7); hs = r >> 3; }
        else { const int j = i - full * HS; rb = full + j / HS; hs = j - (j / HS) * HS; }
    }
    const int n_pairs_all = a.HT / 2;
    const int p0 = hs * (n_pairs_all / HS), p1 = p0 + n_pairs_all / HS;
    const int m0 = (rb * NW + wave) * 16;
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.x3_w);

    // stage g (global counter over the pairs this workgroup walks): its fragments and where they start inside the pair's image
    auto stage_off = [&](int k) -> int {        // k = stage index inside a pair
        if (SINGLE) return 0;
        return k < NS1 ? N2 * G1 * k : N2 * KS + N2 * G2 * (k - NS1);
    };
    auto stage_cnt = [&](int k) -> int {
        if (SINGLE) return CH;
        if (k < NS1) return N2 * (k + 1 < NS1 ? G1 : KS - G1 * (NS1 - 1));
        const int pairs_before = G2 * (k - NS1), tiles = min(KK - 2 * pairs_before, 2 * G2);
        return NT * tiles;
    };
    auto issue = [&](int g) {                   // g counts stages from the first pair of this workgroup
        const int p = p0 + g / NSP, k = g - (g / NSP) * NSP;
        if (p >= p1) return;
        const bf16x8* src = wsrc + ((size_t)p * CH + stage_off(k)) * 64 + lane;
        bf16x8* dst = &x3_wbuf[((g & 1) * SF) * 64];
        const int cnt = stage_cnt(k);
        for (int c = wave; c < cnt; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(dst + c * 64), 16, 0, 0);
    };
    issue(0);
    f32x4 x2sc = {1.f, 1.f, 1.f, 1.f};          // NT = 2: {2^-k1, 2^-k2, 2^k1, 2^k2} of this block's weight image
    float x2sx = 1.f, x2sh = 1.f;               // NT = 2: power-of-two scales of the LayerNorm output / the GELU output (range rule, split_terms.h); folded into x2sc
    if constexpr (NT == 2) {
        x2sc = *reinterpret_cast<const f32x4*>(wsrc + (size_t)n_pairs_all * CH * 64 + 1);
        const f32x4 act = *reinterpret_cast<const f32x4*>(wsrc + (size_t)n_pairs_all * CH * 64 + 2);
        x2sx = act[0]; x2sh = act[1];
    }

    // ---- rows -> k-slot layout of the 32-deep MFMA (lane (row l15, slot group lg) holds channels 32 s + 8 lg .. + 7), LayerNorm in registers, split ----
    const int row = m0 + l15;
    const bool live = row < a.M;
    const float* xr = a.x + (size_t)(live ? row : 0) * CP;
    bf16x8 xs[NT][KS];
    {
        float xv[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = 32 * s + 8 * lg;
            f32x4 v0 = zero4(), v1 = zero4();
            if (c0 < CP) { v0 = ld4(xr + c0); v1 = ld4(xr + c0 + 4); }           // CP % 16 == 0 and c0 % 8 == 0: both halves are inside the row or both outside
            if (!live) { v0 = zero4(); v1 = zero4(); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[s][e] = v0[e]; xv[s][4 + e] = v1[e]; sum += v0[e]; sum += v1[e]; }       // pad channels are exact zeros
        }
        sum = sum_groups(sum);
        const float mean = sum / (float)a.C;
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xv[s][e] - mean; var += d * d; }
        var = sum_groups(var) - (float)(32 * KS - a.C) * mean * mean;             // every zero slot (channel padding and the K padding to 32) added mean^2
        const float rstd = 1.0f / sqrtf(var / (float)a.C + a.eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = 32 * s + 8 * lg;
            f32x4 g0 = zero4(), g1 = zero4(), b0 = zero4(), b1 = zero4();
            if (c0 < CP) { g0 = ld4(a.gamma + c0); g1 = ld4(a.gamma + c0 + 4); b0 = ld4(a.beta + c0); b1 = ld4(a.beta + c0 + 4); }
            float xn[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xn[e] = (xv[s][e] - mean) * rstd * g0[e] + b0[e];                  // gamma = beta = 0 in the pads -> 0
                xn[4 + e] = (xv[s][4 + e] - mean) * rstd * g1[e] + b1[e];
            }
            if constexpr (NT == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xn[e] *= x2sx;
            }
            bf16x8 t[NT];
            split_rows<NT>(xn, t);
#pragma unroll
            for (int i = 0; i < NT; ++i) xs[i][s] = t[i];
        }
    }

    f32x4 acc[KK];
#pragma unroll
    for (int o = 0; o < KK; ++o) acc[o] = zero4();

    int g = 0;                                  // stage counter
    auto next_stage = [&]() -> const bf16x8* {
#if !(ESCX_X3_ABL & 4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // stage g is in LDS for every wave; nobody still reads the other slot
#endif
        issue(g + 1);
        const bf16x8* wb = &x3_wbuf[((g & 1) * SF) * 64 + lane];
        ++g;
        return wb;
    };
    for (int p = p0; p < p1; ++p) {
        const f32x4 bias0 = ld4(a.b1 + 32 * p + 4 * lg), bias1 = ld4(a.b1 + 32 * p + 16 + 4 * lg);
        const bf16x8* wb = nullptr;
        // ---- fc1: two hidden tiles (independent accumulator chains), the bias rides in the accumulator ----
        f32x4 h0 = bias0, h1 = bias1;
        if constexpr (NT == 2) { h0 *= x2sc[2]; h1 *= x2sc[2]; }               // the sums of the scaled weights carry 2^k1: so does the bias (exact)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s % G1 == 0) wb = next_stage();
            const bf16x8* wf = wb + (size_t)((s % G1) * N2) * 64;
            bf16x8 w0[NT], w1[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) { w0[i] = wf[i * 64]; w1[i] = wf[(NT + i) * 64]; }
#define ESCX_X3_FC1(I, J) h0 = mma_x<NT>(w0[I], xs[J][s], h0); h1 = mma_x<NT>(w1[I], xs[J][s], h1);
            if constexpr (NT == 3) { ESCX_X3_TERMS(ESCX_X3_FC1) } else { ESCX_X2_TERMS(ESCX_X3_FC1) }
#undef ESCX_X3_FC1
        }
        // ---- GELU, then the 8 hidden values of this lane become the k-slots of one fc2 step ----
        float hv[8];
#if ESCX_X3_ABL & 2
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[e] = h0[e]; hv[4 + e] = h1[e]; }
        __bf16 s0[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] = (__bf16)hv[e];
        bf16x8 hs3[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) hs3[i] = pack8(s0);
#else
        if constexpr (NT == 2) { h0 *= x2sc[0]; h1 *= x2sc[0]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[e] = gelu_bf(h0[e]); hv[4 + e] = gelu_bf(h1[e]); }
        if constexpr (NT == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] *= x2sh;
        }
        bf16x8 hs3[NT];
        split_rows<NT>(hv, hs3);
#endif
        // ---- fc2: two output tiles per step (no back-to-back MFMAs on one accumulator) ----
        const bf16x8* w2b = SINGLE ? wb + (size_t)(N2 * KS) * 64 : nullptr;
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            if (!SINGLE && op % G2 == 0) w2b = next_stage();
            const int o = 2 * op;
            const bf16x8* wf = w2b + (size_t)((SINGLE ? op : op % G2) * N2) * 64;
            bf16x8 wa[NT], wn[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) { wa[i] = wf[i * 64]; if (o + 1 < KK) wn[i] = wf[(NT + i) * 64]; }
#define ESCX_X3_FC2(I, J) acc[o] = mma_x<NT>(wa[I], hs3[J], acc[o]); if (o + 1 < KK) acc[o + 1] = mma_x<NT>(wn[I], hs3[J], acc[o + 1]);
            if constexpr (NT == 3) { ESCX_X3_TERMS(ESCX_X3_FC2) } else { ESCX_X2_TERMS(ESCX_X3_FC2) }
#undef ESCX_X3_FC2
        }
    }
    if constexpr (NT == 2) {
#pragma unroll
        for (int o = 0; o < KK; ++o) acc[o] *= x2sc[1];       // back from the 2^k2 of the scaled fc2 weights (exact)
    }

    if constexpr (SPLIT) {
        constexpr int KSY = (KK + 1) / 2, TFY = 3 * KSY, UTY = SF / TFY;           // fragments per output tile of the split weights, output tiles per LDS stage
        static_assert(UTY >= 1, "one output tile of the split weights must fit a ring slot");
        const bf16x8* ssrc = reinterpret_cast<const bf16x8*>(a.sp_wf);
        const int n_st = (a.sp_NT + UTY - 1) / UTY;
        auto issue_split = [&](int st, int buf) {
            const int cnt = min(UTY, a.sp_NT - st * UTY) * TFY;
            const bf16x8* src = ssrc + (size_t)(st * UTY) * TFY * 64 + lane;
            for (int c = wave; c < cnt; c += NW)
                __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&x3_wbuf[(buf * SF + c) * 64]), 16, 0, 0);
        };
        __syncthreads();                        // every wave has finished reading the last hidden stage
        issue_split(0, g & 1);
        // y = x + (mlp + b2), LayerNorm (rowgemm_fused_kernel's expressions), split: lane (row l15, channels 16 o + 4 lg + r) - two accumulator tiles = the 8 k-slots of one step
        const float* xres = a.x + (size_t)(live ? row : 0) * CP + 4 * lg;
        float sum = 0.f;
#pragma unroll
        for (int o = 0; o < KK; ++o) {
            const f32x4 res = ld4(xres + 16 * o);
            acc[o] += ld4(a.b2 + 16 * o + 4 * lg);
            acc[o] = live ? res + acc[o] : zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc[o][e];
        }
        sum = sum_groups(sum);
        const float smean = sum / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < KK; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[o][e] - smean; v += d * d; }
        v = sum_groups(v) - (float)(CP - a.C) * smean * smean;
        const float srstd = 1.0f / sqrtf(v / (float)a.C + a.eps);
        bf16x8 ys[3][KSY];
#pragma unroll
        for (int sy = 0; sy < KSY; ++sy) {
            float yn[8];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int o = 2 * sy + hlf;
                if (o < KK) {
                    const f32x4 gm = ld4(a.sp_gamma + 16 * o + 4 * lg), bb = ld4(a.sp_beta + 16 * o + 4 * lg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) yn[4 * hlf + e] = (acc[o][e] - smean) * srstd * gm[e] + bb[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) yn[4 * hlf + e] = 0.f;
                }
            }
            __bf16 t0[8], t1[8], t2[8];
            split3_bf16(yn, t0, t1, t2);
            ys[0][sy] = pack8(t0); ys[1][sy] = pack8(t1); ys[2][sy] = pack8(t2);
        }
        size_t ob0 = 0, ob1 = 0;
        if (live) {
            const int bb = row / (a.sp_H * a.sp_W); const int r0 = row - bb * a.sp_H * a.sp_W; const int hh = r0 / a.sp_W, ww = r0 - hh * a.sp_W;
            ob0 = ((size_t)(bb * 2 * a.sp_H + 2 * hh) * a.sp_W + ww) * a.sp_C2p;
            ob1 = ((size_t)(bb * 2 * a.sp_H + 2 * hh + 1) * a.sp_W + ww) * a.sp_C2p;
        }
        for (int st = 0; st < n_st; ++st) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int buf = (g + st) & 1;
            if (st + 1 < n_st) issue_split(st + 1, buf ^ 1);
            const bf16x8* wbs = &x3_wbuf[(buf * SF) * 64 + lane];
            const int nt_end = min(a.sp_NT, (st + 1) * UTY);
            for (int nt = st * UTY; nt < nt_end; ++nt, wbs += TFY * 64) {
                f32x4 o0 = zero4(), o1 = zero4();
#pragma unroll
                for (int sy = 0; sy < KSY; ++sy) {
                    bf16x8 w[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) w[i] = wbs[(sy * 3 + i) * 64];
                    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], ys[2][sy], o0, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], ys[0][sy], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], ys[1][sy], o0, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], ys[1][sy], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], ys[0][sy], o0, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], ys[0][sy], o1, 0, 0, 0);
                }
                if (live) {
                    const int n = 16 * nt + 4 * lg, s2 = n / a.sp_C2p;
                    st4(a.sp_out + (s2 ? ob1 : ob0) + (n - s2 * a.sp_C2p), o0 + o1);
                }
            }
        }
        return;
    }
    if (!live) return;
    if (HS > 1) {               // raw fc2 partial sums; bias + residual are applied by rows_combine_kernel
        float* pr = a.partial + ((size_t)hs * a.M + row) * CP + 4 * lg;
#pragma unroll
        for (int o = 0; o < KK; ++o) st4(pr + 16 * o, acc[o]);
        return;
    }
    // ---- bias + residual, 16 B per lane (accumulator layout of the 16x16 tile: lane (row l15, channels 16 o + 4 lg .. + 3)) ----
    const float* xres = a.x + (size_t)row * CP + 4 * lg;
    float* orow = (a.out ? a.out : a.x) + (size_t)row * CP + 4 * lg;
    f32x4 res[KK];
#pragma unroll
    for (int o = 0; o < KK; ++o) { res[o] = ld4(xres + 16 * o); acc[o] += ld4(a.b2 + 16 * o + 4 * lg); }
#pragma unroll
    for (int o = 0; o < KK; ++o) st4(orow + 16 * o, res[o] + acc[o]);
}

#ifdef ESCX_EXPERIMENTAL
// Two (TM) 16-row tiles per wave (round 5 experiment, tagged builds only; profiles/r5_mlp_x3_ablation.txt): the wave holds the split rows of TM tiles in registers and
// each weight fragment read from LDS feeds TM MFMAs - the same per-row arithmetic in the same order (bit-identical to mlp_x3_kernel: hash 05247df3f135c41c), half the LDS
// reads, weight DMA pieces and stage barriers per row, at half the resident waves (159 / 228 / 192 registers at C = 45 / 72 / 96).  MEASURED SLOWER: isolated C = 45
// 1.208 -> 1.468 ms per step, C = 72 0.649 -> 0.715, C = 96 0.657 -> 0.870 - the kernel's floor is latency hidden by resident waves, not the fragment stream.
template <int CP, int NW, int TM>
__global__ __launch_bounds__(64 * NW, (mlp_x3_min_waves<CP>() >= 2 * TM ? mlp_x3_min_waves<CP>() / TM : 1)) void mlp_x3_rows_kernel(MlpArgs a) {
    constexpr int KS = mlp_x3_ks(CP), KK = CP / 16, CH = mlp_x3_frags(CP), NOP = (KK + 1) / 2;
    constexpr bool SINGLE = mlp_x3_single(CP);
    constexpr int G1 = mlp_x3_g1(CP), G2 = mlp_x3_g2(CP), NS1 = (KS + G1 - 1) / G1, NS2 = (NOP + G2 - 1) / G2;
    constexpr int NSP = SINGLE ? 1 : NS1 + NS2;
    constexpr int SF = mlp_x3_stage_frags(CP);
    extern __shared__ __attribute__((aligned(16))) bf16x8 x3_wbuf[];             // [2][SF * 64]
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p1 = a.HT / 2;
    const int m0 = (blockIdx.x * NW + wave) * 16 * TM;
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.x3_w);
    auto stage_off = [&](int k) -> int { if (SINGLE) return 0; return k < NS1 ? 6 * G1 * k : 6 * KS + 6 * G2 * (k - NS1); };
    auto stage_cnt = [&](int k) -> int {
        if (SINGLE) return CH;
        if (k < NS1) return 6 * (k + 1 < NS1 ? G1 : KS - G1 * (NS1 - 1));
        const int pairs_before = G2 * (k - NS1), tiles = min(KK - 2 * pairs_before, 2 * G2);
        return 3 * tiles;
    };
    auto issue = [&](int g) {
        const int p = g / NSP, k = g - (g / NSP) * NSP;
        if (p >= p1) return;
        const bf16x8* src = wsrc + ((size_t)p * CH + stage_off(k)) * 64 + lane;
        bf16x8* dst = &x3_wbuf[((g & 1) * SF) * 64];
        const int cnt = stage_cnt(k);
        for (int c = wave; c < cnt; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(dst + c * 64), 16, 0, 0);
    };
    issue(0);

    bf16x8 xs[TM][3][KS];
    bool live[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {              // the expressions of mlp_x3_kernel, per row tile
        const int row = m0 + 16 * t + l15;
        live[t] = row < a.M;
        const float* xr = a.x + (size_t)(live[t] ? row : 0) * CP;
        float xv[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = 32 * s + 8 * lg;
            f32x4 v0 = zero4(), v1 = zero4();
            if (c0 < CP) { v0 = ld4(xr + c0); v1 = ld4(xr + c0 + 4); }
            if (!live[t]) { v0 = zero4(); v1 = zero4(); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[s][e] = v0[e]; xv[s][4 + e] = v1[e]; sum += v0[e]; sum += v1[e]; }
        }
        sum = sum_groups(sum);
        const float mean = sum / (float)a.C;
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xv[s][e] - mean; var += d * d; }
        var = sum_groups(var) - (float)(32 * KS - a.C) * mean * mean;
        const float rstd = 1.0f / sqrtf(var / (float)a.C + a.eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = 32 * s + 8 * lg;
            f32x4 g0 = zero4(), g1 = zero4(), b0 = zero4(), b1 = zero4();
            if (c0 < CP) { g0 = ld4(a.gamma + c0); g1 = ld4(a.gamma + c0 + 4); b0 = ld4(a.beta + c0); b1 = ld4(a.beta + c0 + 4); }
            float xn[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xn[e] = (xv[s][e] - mean) * rstd * g0[e] + b0[e];
                xn[4 + e] = (xv[s][4 + e] - mean) * rstd * g1[e] + b1[e];
            }
            __bf16 s0[8], s1[8], s2[8];
            split3_bf16(xn, s0, s1, s2);
            xs[t][0][s] = pack8(s0); xs[t][1][s] = pack8(s1); xs[t][2][s] = pack8(s2);
        }
    }

    f32x4 acc[TM][KK];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int o = 0; o < KK; ++o) acc[t][o] = zero4();

    int g = 0;
    auto next_stage = [&]() -> const bf16x8* {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(g + 1);
        const bf16x8* wb = &x3_wbuf[((g & 1) * SF) * 64 + lane];
        ++g;
        return wb;
    };
    for (int p = 0; p < p1; ++p) {
        const f32x4 bias0 = ld4(a.b1 + 32 * p + 4 * lg), bias1 = ld4(a.b1 + 32 * p + 16 + 4 * lg);
        const bf16x8* wb = nullptr;
        f32x4 h0[TM], h1[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) { h0[t] = bias0; h1[t] = bias1; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s % G1 == 0) wb = next_stage();
            const bf16x8* wf = wb + (size_t)((s % G1) * 6) * 64;
            bf16x8 w0[3], w1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { w0[i] = wf[i * 64]; w1[i] = wf[(3 + i) * 64]; }
#define ESCX_X3_FC1(I, J) _Pragma("unroll") for (int t = 0; t < TM; ++t) { h0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[I], xs[t][J][s], h0[t], 0, 0, 0); h1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[I], xs[t][J][s], h1[t], 0, 0, 0); }
            ESCX_X3_TERMS(ESCX_X3_FC1)
#undef ESCX_X3_FC1
        }
        bf16x8 hs3[TM][3];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            float hv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { hv[e] = gelu_bf(h0[t][e]); hv[4 + e] = gelu_bf(h1[t][e]); }
            __bf16 s0[8], s1[8], s2[8];
            split3_bf16(hv, s0, s1, s2);
            hs3[t][0] = pack8(s0); hs3[t][1] = pack8(s1); hs3[t][2] = pack8(s2);
        }
        const bf16x8* w2b = SINGLE ? wb + (size_t)(6 * KS) * 64 : nullptr;
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            if (!SINGLE && op % G2 == 0) w2b = next_stage();
            const int o = 2 * op;
            const bf16x8* wf = w2b + (size_t)((SINGLE ? op : op % G2) * 6) * 64;
            bf16x8 wa[3], wn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { wa[i] = wf[i * 64]; if (o + 1 < KK) wn[i] = wf[(3 + i) * 64]; }
#define ESCX_X3_FC2(I, J) _Pragma("unroll") for (int t = 0; t < TM; ++t) { acc[t][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[I], hs3[t][J], acc[t][o], 0, 0, 0); if (o + 1 < KK) acc[t][o + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[I], hs3[t][J], acc[t][o + 1], 0, 0, 0); }
            ESCX_X3_TERMS(ESCX_X3_FC2)
#undef ESCX_X3_FC2
        }
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (!live[t]) continue;
        const int row = m0 + 16 * t + l15;
        const float* xres = a.x + (size_t)row * CP + 4 * lg;
        float* orow = (a.out ? a.out : a.x) + (size_t)row * CP + 4 * lg;
        f32x4 res[KK];
#pragma unroll
        for (int o = 0; o < KK; ++o) { res[o] = ld4(xres + 16 * o); acc[t][o] += ld4(a.b2 + 16 * o + 4 * lg); }
#pragma unroll
        for (int o = 0; o < KK; ++o) st4(orow + 16 * o, res[o] + acc[t][o]);
    }
}

#endif  // ESCX_EXPERIMENTAL

}  // namespace escx
