// Fused backward of the Swin block's MLP for the wide token maps of the training step (C = 45: 19 200 tokens per 3 s clip).
//
//   x2 = x1 + W2 gelu(W1 LN2(x1) + b1) + b2          (attention.py:267-272 inside attention.py:168-176)
//
// The unfused backward (train.hip) is four launches that are HBM-bound on hidden-sized tensors: the forward writes h_pre and gelu(h_pre)
// (tokens x 4C each), dW_fc2 reads the activation, dX_fc2 reads h_pre and writes d h_pre, dW_fc1 and dX_fc1 read d h_pre - seven passes over
// a 530 MB tensor per block at 36 clips.  Here NOTHING hidden-sized exists in memory: the forward is the inference path's fused kernel
// (fused_mlp.h: only x1 stays on the tape) and this kernel recomputes the hidden tile from x1 on the fly.
//
// Work split (one workgroup = NW waves = the NW hidden tiles of 16 units; workgroups are persistent and stride over 16-row tiles):
//   * wave w OWNS hidden tile w: its slices of W1 (two operand layouts) and W2 live in registers for the whole kernel, and so do its
//     accumulators of dW1[16w..16w+15][:] and dW2[:][16w..16w+15] - the weight gradients are contracted over ALL rows the workgroup
//     visits without ever leaving the register file;
//   * per row tile every wave computes, for its hidden tile:  h_pre = xn W1^T + b1,  d h_act = dy W2,  d h_pre = d h_act * gelu'(h_pre),
//     dW1 += d h_pre^T xn,  dW2 += dy^T gelu(h_pre),  and its share of  d xn = d h_pre W1  (16 x C partial sums);
//   * the NW partial sums of d xn meet in LDS and ONE wave (the role rotates) adds them in wave order, applies the LayerNorm backward
//     and writes dx1 = dy + LN2'(d xn) (token order, and window-slot order for the attention backward).  That wave is on the critical path of
//     the iteration (everybody meets it at the next barrier), so it does nothing else: the LayerNorm's parameter gradients are NOT column
//     sums over the rows here (a first version did that with 108 cross-lane shuffles per tile: 1.69 ms per launch, 33 TFLOP/s) but algebra:
//     with E = d h_pre^T xhat (what the wave accumulates instead of dW1; xhat = the normalised rows before the affine map),
//        dW1 = E diag(gamma) + d b1 beta^T,    d gamma = colsum(W1 * E),    d beta = W1^T d b1      (mlp_bwd_finish_kernel, exact identities);
//   * the row tile (LayerNorm applied once) is staged in LDS by two other waves one iteration ahead, in both operand layouts
//     (row-major for the contractions over channels, transposed for the contractions over rows), so no wave ever waits on global memory.
// One barrier per row tile.  No atomics: per-workgroup partial sums of every parameter gradient are reduced in a fixed order by the caller
// (reduce_partials), so the step stays run-to-run deterministic.
//
// MFMA operand conventions (v_mfma_f32_16x16x4_f32, lane = (g = lane >> 4, b = lane & 15)): A-operand register = A[i = b][k-slot g],
// B-operand register = B[k-slot g][j = b], D register r = D[i = 4g + r][j = b].  A D tile is therefore directly the B operand of a
// contraction over its ROW index (k-slot g <-> rows 4g + r at step r); a contraction over its COLUMN index needs the transposed tile
// (one 16 x 16 transpose of d h_pre per tile through a private LDS scratch).
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"
#include "train_kernels.h"
#include "split_terms.h"

namespace escx {

struct MlpBwdArgs {
    const float* x1;            // [M][CP]   input of LN2 (kept on the tape)
    const float* dy;            // [M][CP]   gradient w.r.t. the block output x2
    float* dx1;                 // [M][CP]   out: dy + LN2 backward
    float* dx1s;                // [B*slots][CP] the same rows in window-slot order (nullptr: not written)
    const int* slot_of;         // token -> slot (per clip)
    const float* gamma; const float* beta;      // LN2 [CP]
    const float* w1;            // [hiddenP][CP]
    const float* b1;            // [hiddenP]
    const float* w2T;           // [hiddenP][CP]   (w2T[h][c] = W2[c][h])
    const float* w1T;           // [CP][hiddenP]
    float* part;                // per-workgroup partial sums, regions [grid.x][n]: E = d h_pre^T xhat (hiddenP*CP) | dW2 (CP*hiddenP) | db1 (hiddenP) | db2 (CP)
    float* dxn_part;            // hidden split (grid.y = HS > 1): slab [HS][M][CP] of d xn partial sums; the LayerNorm backward then runs as a separate pass
    int M, C, hiddenP, tokens, slots;
    float eps;
    int dbg;                    // timing experiments only (ESCX_MLPBWD_DBG): 1 = finisher idle, 2 = stagers stage only the first tile, 4 = compute waves skip the partial-sum writes
};

// NC compute waves (= hidden tiles) + 3 service waves: wave NC stages x1 (LayerNorm), wave NC + 1 stages dy, wave NC + 2 finishes the PREVIOUS
// tile (cross-wave sum of the d xn partials, LayerNorm backward, stores).  A first version rotated these roles over the compute waves: with
// one barrier per tile everybody then waits for the wave that had the extra role (0.92 ms per launch); with dedicated waves the compute waves
// all do the same work between two barriers and a service wave has a whole iteration (~6000 cycles) for a few hundred cycles of work.
// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)): the staging waves
// would then wait at every tile for the global loads they have just issued for the tile after next (HBM latency per iteration: 745 us per launch
// instead of ~480), and the finishing wave for its stores.  The data exchanged between waves here lives in LDS; global loads stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// HS > 1 (grid.y = HS; C = 72: 18 hidden tiles do not fit the 16 waves of a workgroup): workgroup (x, y) owns hidden tiles [y * NC, (y + 1) * NC); its d xn is
// then a partial sum over a part of the hidden units, so the finisher only adds its NC waves' shares and writes them to slab y - the caller sums the HS slabs and runs
// the stand-alone LayerNorm backward.  The finisher then reads nothing of the staged tile, the ring needs two slots instead of three, and the normalised rows are not
// staged row-major (LDS: 156 KB at C = 72 with NC = 9).  W1's second operand layout is re-read from L1 per tile there instead of living in 20 registers.
// X2 (round 6, third session): the two contractions over the CHANNELS - h_pre = xn W1^T and d h_act = dy W2, 2 x KC x 4 fp32 MFMAs of 32 cycles per row tile and wave - run
// on v_mfma_f32_16x16x32_f16 with both operands split into two fp16 terms and three cross products (split_terms.h NT = 2): 2 x KS x 3 instructions of ~17 cycles.  The range
// rule holds by construction, with powers of two only: a wave scales ITS weight tiles by their own maximum (once per launch; a hidden tile's columns are independent, so a
// per-tile scale is exact), the x1 stager scales the LayerNorm output by the bound max |gamma| sqrt(C) + max |beta|, the dy stager scales every 16-row gradient tile by the
// tile's own maximum (block floating point: gradients have no a-priori bound); the accumulators are multiplied back by the inverse powers of two.  The stagers split ONCE per
// row tile for all the compute waves and write the terms in MFMA fragment order ([k step][term][lane][8 halves]: conflict-free ds_read_b128), instead of the fp32 rows.
// The three contractions over rows / hidden units (K = 16) stay on the fp32 MFMA: a 32-deep step would be half empty and their operands are produced per wave.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4_f16(const f32x4 v, float scale, uint2& hi, uint2& lo) {
#pragma clang fp contract(off)
    half4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float x = v[e] * scale; const _Float16 x1 = (_Float16)x; h[e] = x1; l[e] = (_Float16)(x - (float)x1); }
    hi = __builtin_bit_cast(uint2, h); lo = __builtin_bit_cast(uint2, l);
}
__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

template <int CP, int NC, int HS, bool X2 = false>
__global__ __launch_bounds__(64 * (NC + 3)) void mlp_bwd_fused_kernel(MlpBwdArgs a) {
    constexpr int KC = CP / 16;                 // channel tiles
    constexpr int KS = (CP + 31) / 32;          // X2: 32-deep k steps
    constexpr int SLD = CP + 4, TLD = 20;       // row strides (dwords) of the row-major / transposed staged copies
    constexpr int NSTG = HS > 1 ? 2 : 3;
    constexpr bool W1E_REG = CP <= 48 && !X2;   // X2: the split weight fragments take the registers (the second layout of W1 is re-read from L1 per tile, as at C = 72)
    struct Stage {
        bf16x8 xnp[X2 ? KS * 2 * 64 : 1], dyp[X2 ? KS * 2 * 64 : 1];      // X2: split terms of the LayerNorm output / of dy, fragment order
        float xn[X2 ? 4 : 16 * SLD]; float xh[HS > 1 ? 4 : 16 * SLD]; float dy[(X2 && HS > 1) ? 4 : 16 * SLD]; float xhT[CP * TLD]; float dyT[CP * TLD]; float rstd[16];
        float inv_dy[4];                                                  // X2: 1 / (power-of-two scale of this tile's dy)
    };
    __shared__ Stage stg[NSTG];
    __shared__ float sx_s[2];                   // X2: power-of-two scale of the LayerNorm output and its inverse
    __shared__ float red[2][NC][16 * SLD];
    __shared__ float tr[NC][16 * TLD];
    __shared__ float gam_s[CP], bet_s[CP];

    const int lane = threadIdx.x & 63, b = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = (a.M + 15) / 16;
    const int n_it = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const float invC = 1.0f / (float)a.C;
    if (threadIdx.x < CP) { gam_s[threadIdx.x] = a.gamma[threadIdx.x]; bet_s[threadIdx.x] = a.beta[threadIdx.x]; }
    __syncthreads();
    if constexpr (X2) {
        if (threadIdx.x == 0) { const float sx = act_pow2_scale(ln_out_bound(gam_s, bet_s, CP, a.C)); sx_s[0] = sx; sx_s[1] = 1.0f / sx; }
        __syncthreads();
    }
    // X2: where the 4 channels (16 ct + 4 g ..) of row b go in fragment order: k step ct / 2, lane (b, k group 2 (ct & 1) + g / 2), halves 4 (g & 1) .. + 3
    auto frag_slot = [&](int ct, int term) { return (((ct >> 1) * 2 + term) * 64 + (2 * (ct & 1) + (g >> 1)) * 16 + b) * 2 + (g & 1); };
    auto tile_row = [&](int t) { return ((int)blockIdx.x + t * (int)gridDim.x) * 16 + b; };
    auto ring_next = [&](int i) { return i + 1 == NSTG ? 0 : i + 1; };

    // The service waves are the youngest of the workgroup: at equal priority the SIMD's arbiter hands them only the issue slots the three compute
    // waves leave over, and their short per-tile chains then take longer than the compute waves' tile (measured: the finisher +120 us, the stagers
    // +73 us per launch on a 530 us compute loop).  They issue a few hundred instructions per tile, so they go first.
    if (wave >= NC) __builtin_amdgcn_s_setprio(3);
    if (wave == NC) {
        // ---- stager of x1: LayerNorm once per row, both operand layouts; global loads run two tiles ahead of the consumers ----
        f32x4 pre[KC];
        auto load = [&](int t) {
            const int row = tile_row(t);
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) pre[ct] = (t < n_it && row < a.M) ? ld4(a.x1 + (size_t)row * CP + 16 * ct + 4 * g) : zero4();
        };
        auto store = [&](Stage& S, int t) {
            const bool live = tile_row(t) < a.M;
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < KC; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (16 * ct + 4 * g + e < a.C) s += pre[ct][e];
            const float mean = sum_groups(s) * invC;
            float var = 0.f;
#pragma unroll
            for (int ct = 0; ct < KC; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (16 * ct + 4 * g + e < a.C) { const float d = pre[ct][e] - mean; var += d * d; }
            const float rstd = 1.0f / sqrtf(sum_groups(var) * invC + a.eps);
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) {
                const f32x4 bt = ld4(&bet_s[16 * ct + 4 * g]), gm = ld4(&gam_s[16 * ct + 4 * g]);
                f32x4 xh, xn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = live && (16 * ct + 4 * g + e < a.C);
                    xh[e] = in ? (pre[ct][e] - mean) * rstd : 0.f;
                    xn[e] = in ? xh[e] * gm[e] + bt[e] : 0.f;
                    S.xhT[(16 * ct + 4 * g + e) * TLD + b] = xh[e];
                }
                if (HS == 1) st4(&S.xh[b * SLD + 16 * ct + 4 * g], xh);
                if constexpr (X2) {
                    uint2 hi, lo;
                    split4_f16(xn, sx_s[0], hi, lo);
                    reinterpret_cast<uint2*>(S.xnp)[frag_slot(ct, 0)] = hi; reinterpret_cast<uint2*>(S.xnp)[frag_slot(ct, 1)] = lo;
                } else st4(&S.xn[b * SLD + 16 * ct + 4 * g], xn);
            }
            if constexpr (X2) {                  // K padding of the last 32-deep step: zero halves
#pragma unroll
                for (int ct = KC; ct < 2 * KS; ++ct) { reinterpret_cast<uint2*>(S.xnp)[frag_slot(ct, 0)] = uint2{0u, 0u}; reinterpret_cast<uint2*>(S.xnp)[frag_slot(ct, 1)] = uint2{0u, 0u}; }
            }
            if (g == 0) S.rstd[b] = live ? rstd : 0.f;
        };
        load(0); store(stg[0], 0); load(1);
        lds_barrier();
        int snext = 1;
        for (int it = 0; it < n_it; ++it) {
            if (it + 1 < n_it && !(a.dbg & 2)) { store(stg[snext], it + 1); load(it + 2); }
            snext = ring_next(snext);
            lds_barrier();
        }
    } else if (wave == NC + 1) {
        // ---- stager of dy ----
        f32x4 pre[KC];
        auto load = [&](int t) {
            const int row = tile_row(t);
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) pre[ct] = (t < n_it && row < a.M) ? ld4(a.dy + (size_t)row * CP + 16 * ct + 4 * g) : zero4();
        };
        f32x4 sdy[KC];                           // d b2 = column sums of dy: this lane's row of every tile (rows beyond M are loaded as zeros)
#pragma unroll
        for (int ct = 0; ct < KC; ++ct) sdy[ct] = zero4();
        auto store = [&](Stage& S) {
            float dsc = 1.0f;
            if constexpr (X2) {                  // block floating point: this tile's own maximum -> power of two (1 for an all-zero tile)
                float m = 0.f;
#pragma unroll
                for (int ct = 0; ct < KC; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(pre[ct][e]));
                dsc = x2_scale(__float_as_uint(wave_max(m)));
                if (lane == 0) S.inv_dy[0] = 1.0f / dsc;
            }
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) {
                sdy[ct] += pre[ct];
                if constexpr (!(X2 && HS > 1)) st4(&S.dy[b * SLD + 16 * ct + 4 * g], pre[ct]);
                if constexpr (X2) {
                    uint2 hi, lo;
                    split4_f16(pre[ct], dsc, hi, lo);
                    reinterpret_cast<uint2*>(S.dyp)[frag_slot(ct, 0)] = hi; reinterpret_cast<uint2*>(S.dyp)[frag_slot(ct, 1)] = lo;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) S.dyT[(16 * ct + 4 * g + e) * TLD + b] = pre[ct][e];
            }
            if constexpr (X2) {
#pragma unroll
                for (int ct = KC; ct < 2 * KS; ++ct) { reinterpret_cast<uint2*>(S.dyp)[frag_slot(ct, 0)] = uint2{0u, 0u}; reinterpret_cast<uint2*>(S.dyp)[frag_slot(ct, 1)] = uint2{0u, 0u}; }
            }
        };
        load(0); store(stg[0]); load(1);
        lds_barrier();
        int snext = 1;
        for (int it = 0; it < n_it; ++it) {
            if (it + 1 < n_it && !(a.dbg & 2)) { store(stg[snext]); load(it + 2); }
            snext = ring_next(snext);
            lds_barrier();
        }
        float* Pd = a.part + (size_t)gridDim.x * (2 * (size_t)a.hiddenP * CP + a.hiddenP) + (size_t)blockIdx.x * CP;
#pragma unroll
        for (int ct = 0; ct < KC; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = sdy[ct][e];
#pragma unroll
                for (int sh = 8; sh >= 1; sh >>= 1) u += __shfl_xor(u, sh, 16);
                if (b == 0 && blockIdx.y == 0) Pd[16 * ct + 4 * g + e] = u;
            }
    } else if (wave == NC + 2) {
        // ---- finisher: during iteration it it completes tile it - 1 (the partial sums were published by the barrier that ended it - 1) ----
        auto finish = [&](int t, const Stage& S) {
            const int row = tile_row(t);
            const bool live = row < a.M;
            const float* rp = &red[t & 1][0][0];
            if constexpr (HS > 1) {
                // partial sum over this workgroup's hidden tiles -> slab blockIdx.y (the other workgroups of the row block hold the rest)
                float* dst = a.dxn_part + ((size_t)blockIdx.y * a.M + row) * CP;
#pragma unroll
                for (int ct = 0; ct < KC; ++ct) {
                    const int o4 = b * SLD + 16 * ct + 4 * g;
                    f32x4 tt = ld4(rp + o4);
#pragma unroll
                    for (int w = 1; w < NC; ++w) tt += ld4(rp + w * 16 * SLD + o4);           // wave order: fixed
                    if (live) st4(dst + 16 * ct + 4 * g, tt);
                }
                return;
            }
            f32x4 gv[KC], xh[KC];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) {
                const int o4 = b * SLD + 16 * ct + 4 * g;
                f32x4 tt = ld4(rp + o4);
#pragma unroll
                for (int w = 1; w < NC; ++w) tt += ld4(rp + w * 16 * SLD + o4);           // wave order: fixed
                xh[ct] = ld4(&S.xh[o4]);
                tt *= ld4(&gam_s[16 * ct + 4 * g]);                                       // gamma = 0 in the pad channels
                gv[ct] = tt;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += tt[e]; s2 += tt[e] * xh[ct][e]; }
            }
            const float c1 = sum_groups(s1) * invC, c2 = sum_groups(s2) * invC;
            const float rstd = S.rstd[b];
            if (live) {
                float* ds = nullptr;
                if (a.dx1s) { const int bi = row / a.tokens, rr = row - bi * a.tokens; ds = a.dx1s + ((size_t)bi * a.slots + a.slot_of[rr]) * CP; }
#pragma unroll
                for (int ct = 0; ct < KC; ++ct) {
                    const f32x4 dyv = ld4(&S.dy[b * SLD + 16 * ct + 4 * g]);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (16 * ct + 4 * g + e < a.C) ? dyv[e] + rstd * (gv[ct][e] - c1 - xh[ct][e] * c2) : 0.f;
                    st4(a.dx1 + (size_t)row * CP + 16 * ct + 4 * g, o);
                    if (ds) st4(ds + 16 * ct + 4 * g, o);
                }
            }
        };
        lds_barrier();
        int sprev = NSTG - 1;                    // (it - 1) % NSTG (HS > 1: the finisher does not read the stage)
        for (int it = 0; it < n_it; ++it) {
            if (it > 0 && !(a.dbg & 1)) finish(it - 1, stg[sprev]);
            sprev = ring_next(sprev);
            lds_barrier();
        }
        if (n_it > 0) finish(n_it - 1, stg[sprev]);
    } else {
        // ---- compute wave w: hidden tile w ----
        const int ht = (int)blockIdx.y * NC + wave;                      // this wave's hidden tile
        f32x4 W1a[X2 ? 1 : KC], W2c[X2 ? 1 : KC], W1e[W1E_REG ? KC : 1], dW1T[KC], dW2[KC];
        bf16x8 w1s[X2 ? KS : 1][2], w2s[X2 ? KS : 1][2];       // X2: this wave's rows of W1 / W2^T as two fp16 terms, column operand of the 32-deep steps: lane (hidden b, k group g)
        float inv1 = 1.f, inv2 = 1.f;                           // X2: 1 / (weight scale x activation scale), powers of two
#pragma unroll
        for (int ct = 0; ct < KC; ++ct) {
            if constexpr (!X2) {
                W1a[ct] = ld4(a.w1 + (size_t)(16 * ht + b) * CP + 16 * ct + 4 * g);
                W2c[ct] = ld4(a.w2T + (size_t)(16 * ht + b) * CP + 16 * ct + 4 * g);
            }
            if (W1E_REG) W1e[ct] = ld4(a.w1T + (size_t)(16 * ct + b) * a.hiddenP + 16 * ht + 4 * g);
            dW1T[ct] = zero4(); dW2[ct] = zero4();
        }
        if constexpr (X2) {
            float v1[KS][8], v2[KS][8], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int c0 = 32 * s + 8 * g;
                f32x4 p0 = zero4(), p1 = zero4(), q0 = zero4(), q1 = zero4();
                if (c0 < CP) {                                  // CP % 16 == 0 and c0 % 8 == 0: both halves inside the row or both outside
                    const float* r1 = a.w1 + (size_t)(16 * ht + b) * CP + c0; const float* r2 = a.w2T + (size_t)(16 * ht + b) * CP + c0;
                    p0 = ld4(r1); p1 = ld4(r1 + 4); q0 = ld4(r2); q1 = ld4(r2 + 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v1[s][e] = p0[e]; v1[s][4 + e] = p1[e]; v2[s][e] = q0[e]; v2[s][4 + e] = q1[e];
                    m1 = fmaxf(m1, fmaxf(fabsf(p0[e]), fabsf(p1[e]))); m2 = fmaxf(m2, fmaxf(fabsf(q0[e]), fabsf(q1[e])));
                }
            }
            const float s1 = x2_scale(__float_as_uint(wave_max(m1))), s2 = x2_scale(__float_as_uint(wave_max(m2)));
#pragma unroll
            for (int s = 0; s < KS; ++s) { split_terms<2>(v1[s], w1s[s], s1); split_terms<2>(v2[s], w2s[s], s2); }
            inv1 = (1.0f / s1) * sx_s[1]; inv2 = 1.0f / s2;
        }
        const float* w1e_src = a.w1T + (size_t)b * a.hiddenP + 16 * ht + 4 * g;    // + 16 * ct * hiddenP per channel tile
        const float bias1 = a.b1[16 * ht + b];
        float db1 = 0.f;
        float* trw = &tr[wave][0];
        lds_barrier();
        int sidx = 0;
        for (int it = 0; it < n_it; ++it) {
            const Stage& S = stg[sidx];
            f32x4 hp = {bias1, bias1, bias1, bias1}, dh = zero4();
            if constexpr (X2) {
                f32x4 ha = zero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 xh_ = S.xnp[(s * 2 + 0) * 64 + lane], xl_ = S.xnp[(s * 2 + 1) * 64 + lane];
                    const bf16x8 dh_ = S.dyp[(s * 2 + 0) * 64 + lane], dl_ = S.dyp[(s * 2 + 1) * 64 + lane];
                    ha = mma_x<2>(xl_, w1s[s][0], ha); dh = mma_x<2>(dl_, w2s[s][0], dh);            // smallest terms first
                    ha = mma_x<2>(xh_, w1s[s][1], ha); dh = mma_x<2>(dh_, w2s[s][1], dh);
                    ha = mma_x<2>(xh_, w1s[s][0], ha); dh = mma_x<2>(dh_, w2s[s][0], dh);
                }
                const float i2 = inv2 * S.inv_dy[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) { hp[r] = __builtin_fmaf(ha[r], inv1, bias1); dh[r] *= i2; }     // back from the power-of-two scales (exact)
            } else {
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) {
                const f32x4 xa = ld4(&S.xn[b * SLD + 16 * ct + 4 * g]);
                const f32x4 da = ld4(&S.dy[b * SLD + 16 * ct + 4 * g]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hp = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[r], W1a[ct][r], hp, 0, 0, 0);        // h_pre[row 4g+r'][hid b]
                    dh = __builtin_amdgcn_mfma_f32_16x16x4f32(da[r], W2c[ct][r], dh, 0, 0, 0);        // d h_act
                }
            }
            }
            // the row-contraction operands are fetched BEFORE the GELU arithmetic so that their LDS latency hides under it (all waves of the
            // workgroup run in lockstep behind the per-tile barrier: an exposed LDS round trip is paid by every SIMD at the same time)
            constexpr bool PREFETCH_T = KC <= 3;            // wider maps: fetched tile by tile inside the contraction (registers)
            f32x4 xt[PREFETCH_T ? KC : 1], dt[PREFETCH_T ? KC : 1];
            if constexpr (PREFETCH_T) {
#pragma unroll
                for (int ct = 0; ct < KC; ++ct) {
                    xt[ct] = ld4(&S.xhT[(16 * ct + b) * TLD + 4 * g]);
                    dt[ct] = ld4(&S.dyT[(16 * ct + b) * TLD + 4 * g]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 hact, dhp;
#pragma unroll
            for (int r = 0; r < 4; ++r) { float ac, gr; gelu_pair(hp[r], ac, gr); hact[r] = ac; dhp[r] = dh[r] * gr; }
            db1 += (dhp[0] + dhp[1]) + (dhp[2] + dhp[3]);
            // transpose d h_pre through the wave's private scratch: written [hid b][rows 4g..4g+3], read [hid 4g+r][row b]
            st4(trw + b * TLD + 4 * g, dhp);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4 dhT;
#pragma unroll
            for (int r = 0; r < 4; ++r) dhT[r] = trw[(4 * g + r) * TLD + b];
            __builtin_amdgcn_sched_barrier(0);
            // weight gradients: contractions over the 16 rows (k-slot g <-> rows 4g + r)
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) {
                f32x4 xq, dq;
                if constexpr (PREFETCH_T) { xq = xt[ct]; dq = dt[ct]; }
                else { xq = ld4(&S.xhT[(16 * ct + b) * TLD + 4 * g]); dq = ld4(&S.dyT[(16 * ct + b) * TLD + 4 * g]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dW1T[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[r], dhp[r], dW1T[ct], 0, 0, 0);   // E[hid b][c 16ct+4g+r'] = sum_rows d h_pre * xhat
                    dW2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[r], hact[r], dW2[ct], 0, 0, 0);    // dW2[c 16ct+4g+r'][hid b]
                }
            }
            // this tile's share of d xn = d h_pre W1: D[c 16ct+4g+r'][row b]; the KC accumulators are independent chains
            f32x4 dx[KC];
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) dx[ct] = zero4();
            if constexpr (W1E_REG) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < KC; ++ct) dx[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1e[ct][r], dhT[r], dx[ct], 0, 0, 0);
            } else {
                f32x4 we[KC];
#pragma unroll
                for (int ct = 0; ct < KC; ++ct) we[ct] = ld4(w1e_src + (size_t)16 * ct * a.hiddenP);      // L1-resident: the same 5 KB every tile
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < KC; ++ct) dx[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(we[ct][r], dhT[r], dx[ct], 0, 0, 0);
            }
            float* rp = &red[it & 1][wave][0];
#pragma unroll
            for (int ct = 0; ct < KC; ++ct) if (!(a.dbg & 4) || it == 0) st4(rp + b * SLD + 16 * ct + 4 * g, dx[ct]);
            sidx = ring_next(sidx);
            lds_barrier();
        }
        // per-workgroup partial sums of the parameter gradients
        const size_t n1 = (size_t)a.hiddenP * CP;
        float* P1 = a.part + (size_t)blockIdx.x * n1;
        float* P2 = a.part + (size_t)gridDim.x * n1 + (size_t)blockIdx.x * n1;
        float* Pb = a.part + (size_t)gridDim.x * 2 * n1 + (size_t)blockIdx.x * a.hiddenP;
#pragma unroll
        for (int ct = 0; ct < KC; ++ct) {
            st4(P1 + (size_t)(16 * ht + b) * CP + 16 * ct + 4 * g, dW1T[ct]);
#pragma unroll
            for (int r = 0; r < 4; ++r) P2[(size_t)(16 * ct + 4 * g + r) * a.hiddenP + 16 * ht + b] = dW2[ct][r];
        }
        const float sb = sum_groups(db1);
        if (g == 0) Pb[16 * ht + b] = sb;
    }
}

// Fixed-order sum of the per-workgroup partials (slice-major regions E | dW2 | db1 | db2) in ONE launch: thread i owns output element i and adds
// the `slices` values in slice order (consecutive threads read consecutive addresses of a slice: coalesced).  dW2, db1 and db2 are final; E
// and db1 feed mlp_bwd_finish_kernel.
static __global__ __launch_bounds__(256) void mlp_bwd_reduce_kernel(const float* __restrict__ part, int slices, int n1, int hiddenP, int Cp,
                                                                    float* __restrict__ E, float* __restrict__ dW2, float* __restrict__ db1, float* __restrict__ db2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = 2 * n1 + hiddenP + Cp;
    if (i >= total) return;
    const float* src; float* dst; int n, j;
    if (i < n1) { src = part; dst = E; n = n1; j = i; }
    else if (i < 2 * n1) { src = part + (size_t)slices * n1; dst = dW2; n = n1; j = i - n1; }
    else if (i < 2 * n1 + hiddenP) { src = part + (size_t)slices * 2 * n1; dst = db1; n = hiddenP; j = i - 2 * n1; }
    else { src = part + (size_t)slices * (2 * n1 + hiddenP); dst = db2; n = Cp; j = i - 2 * n1 - hiddenP; }
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < slices; ++k) t += src[(size_t)k * n + j];
    dst[j] = t;
}

// d xn = slab 0 + slab 1 + ... (fixed order) of the hidden-split form
static __global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int n_slabs, long long n4, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = ld4(slabs + 4 * i);
    for (int k = 1; k < n_slabs; ++k) v += ld4(slabs + (size_t)k * 4 * n4 + 4 * i);
    st4(out + 4 * i, v);
}

// dW1 = E diag(gamma) + d b1 beta^T ;  d gamma[c] = sum_h W1[h][c] E[h][c] ;  d beta[c] = sum_h W1[h][c] d b1[h]   (fixed summation order)
static __global__ __launch_bounds__(1024) void mlp_bwd_finish_kernel(const float* __restrict__ E, const float* __restrict__ db1, const float* __restrict__ w1,
                                             const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dW1,
                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int hiddenP, int Cp) {
    __shared__ float pg[16][100], pb[16][100];
    for (int i = threadIdx.x; i < hiddenP * Cp; i += blockDim.x) {
        const int hh = i / Cp, c = i - hh * Cp;
        dW1[i] = E[i] * gamma[c] + db1[hh] * beta[c];
    }
    // 16 threads per column, each a strided subset of the hidden units in increasing order; the 16 partial sums are then added in order
    const int c = threadIdx.x % 64, k = threadIdx.x / 64;       // Cp <= 64 columns x 16 parts
    if (!dgamma) return;                        // hidden-split form: the stand-alone LayerNorm backward produces d gamma / d beta
    if (c < Cp) {
        float sg = 0.f, sb = 0.f;
        for (int hh = k; hh < hiddenP; hh += 16) { const float w = w1[(size_t)hh * Cp + c]; sg += w * E[(size_t)hh * Cp + c]; sb += w * db1[hh]; }
        pg[k][c] = sg; pb[k][c] = sb;
    }
    __syncthreads();
    if ((int)threadIdx.x < Cp) {
        float sg = pg[0][threadIdx.x], sb = pb[0][threadIdx.x];
        for (int j = 1; j < 16; ++j) { sg += pg[j][threadIdx.x]; sb += pb[j][threadIdx.x]; }
        dgamma[threadIdx.x] = sg; dbeta[threadIdx.x] = sb;
    }
}

}  // namespace escx
