// Fused Swin MLP for gfx950:  x <- x + W2 . gelu(W1 . LN(x) + b1) + b2     (attention.py:177, 258-272)
//
// Wave-autonomous, register-resident, barrier-free.  One wave owns 16*TM token rows end to end:
//   1. loads its rows straight into the MFMA *operand* layout (lane (row = l&15, k-slot group g = l>>4)
//      holds x[row][16kk + 4g .. +3]) and LayerNorms them in registers (row statistics are an in-lane sum
//      plus two cross-lane shuffles over the 4 k-slot groups);
//   2. walks the hidden dimension 16 units at a time: fc1 tile -> bias -> exact-erf GELU -> fc2 partial.
//      The fc1 accumulator tile D[n][m] leaves lane (m = l&15, g) holding h[m][16t + 4g + r], r = 0..3,
//      which is *already* the operand layout of the fc2 MFMA with k-slot (g, r) <-> hidden unit 16t+4g+r.
//      The tokens x 4C hidden activation therefore never exists anywhere but in 4*TM registers;
//   3. adds bias + residual and stores 16 bytes per lane.
// Weights are pre-packed on the host in fragment order ([tile][k-step][lane][4]) so that every weight
// fetch is one fully coalesced 1 KiB wave load from L2 (all waves walk the same 2 x 4C x C floats, which
// stay L2-resident); no LDS staging is needed because a fragment is consumed by exactly one wave-level
// MFMA sequence and TM row tiles amortise it inside the wave.
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32).  Algorithmic traffic: read x once, write x once.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"
#include "launchers.h"

namespace escx {

struct MlpArgs {
    float* x;                   // [M][CP] in/out
    const float* gamma; const float* beta;      // LayerNorm (norm2), padded to CP
    const f32x4* w1f; const float* b1;          // fc1 fragments [HT][KK][64], bias [16*HT]
    const f32x4* w2f; const float* b2;          // fc2 fragments [KK][HT][64], bias [CP]
    const f32x4* wcf;                           // per hidden tile: [HT][KK fc1 fragments | KK fc2 fragments][64] (LDS-staged variant)
    int M, C, HT;
    float eps;
    unsigned long long* trace;                  // debug (ABL bit 64): per-wave cycle sums of the main-loop phases
    // Hidden split (LDS-staged kernel only): with fewer row tiles than SIMDs (deep layers at small batch) the grid is
    // row-blocks x HS and workgroup (rb, hs) walks hidden tiles [hs*HT/HS, (hs+1)*HT/HS); the fc2 partial sums go to
    // partial[hs][M][CP] and mlp_combine_kernel adds them in fixed order (deterministic, no atomics).
    int HS; float* partial;
    float* out;                 // nullptr: in place; otherwise x is left untouched and x + mlp(x) goes to out (training forward: x1 stays on the tape)
    // Hidden split with the combine INSIDE the launch (round 4): one arrival counter per row block (zero between launches).  Every workgroup
    // publishes its slab write-through and takes a ticket; the LAST arriver adds the slabs in slab-index order - ((P0 + P1) + P2) + bias, then
    // + x, the arithmetic of rows_combine_kernel - and writes x.  nullptr: slabs only, the caller runs rows_combine_kernel.
    int* tickets;
    // PatchSplit in the epilogue (SPLIT instantiations, round 5; scale.py:131-145): the wave still owns its 16 rows x + mlp(x) in registers after the
    // residual add, and nothing else reads the pre-split map (csrvq.py:173-181), so LayerNorm(C) -> Linear(C -> 2 C', no bias) -> two-row scatter
    // runs right there: the split weights (fragment order, Layer::sub_wf) stream through the same LDS ring as further stages, two output tiles per
    // stage.  The arithmetic is rowgemm_fused_kernel's (statistics in lane order then over the 4 k-slot groups, one k-ordered chain per output).
    // x is NOT written.  sp_wf == nullptr: plain epilogue.
    const f32x4* sp_wf; const float* sp_gamma; const float* sp_beta; float* sp_out;
    int sp_NT, sp_H, sp_W, sp_C2p;
    const void* x3_w;           // fused_mlp_x3.h: split (3 x bf16) weight image [pair of hidden tiles][fragment][lane][8 bf16], or nullptr
};

// buffer descriptor over [p, p + bytes): raw (stride 0) addressing; for write-through (sc1) stores of hand-off data
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)bytes, 0x00020000);
}

template <int CP, int TM>
__global__ __launch_bounds__(256) void mlp_fused_kernel(MlpArgs a) {
    constexpr int KK = CP / 16;
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int m0 = wave * (16 * TM);
    if (m0 >= a.M) return;

    // ---- 1. rows -> operand layout, LayerNorm in registers -------------------------------------
    f32x4 xf[TM][KK];
    float mean[TM], rstd[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = m0 + t * 16 + l15;
        const float* xr = a.x + (size_t)row * CP + 4 * lg;
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            xf[t][kk] = row < a.M ? ld4(xr + 16 * kk) : zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e) s += xf[t][kk][e];            // pad channels are exact zeros
        }
        s = sum_groups(s);
        mean[t] = s / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xf[t][kk][e] - mean[t]; v += d * d; }
        v = sum_groups(v) - (float)(CP - a.C) * mean[t] * mean[t];       // the zero pads each added mean^2
        rstd[t] = 1.0f / sqrtf(v / (float)a.C + a.eps);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const f32x4 g = ld4(a.gamma + 16 * kk + 4 * lg), b = ld4(a.beta + 16 * kk + 4 * lg);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xf[t][kk][e] = (xf[t][kk][e] - mean[t]) * rstd[t] * g[e] + b[e];    // gamma = beta = 0 in the pads -> 0
    }

    // ---- 2. hidden tiles: fc1 -> GELU -> fc2 partial, all in registers ---------------------------
    f32x4 acc[KK][TM];
#pragma unroll
    for (int o = 0; o < KK; ++o)
#pragma unroll
        for (int t = 0; t < TM; ++t) acc[o][t] = zero4();

    const f32x4* w1 = a.w1f + lane;
    const f32x4* w2 = a.w2f + lane;
    for (int ht = 0; ht < a.HT; ++ht) {
        f32x4 h[TM], h2[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) { h[t] = zero4(); h2[t] = zero4(); }
        const f32x4* w1t = w1 + (size_t)ht * KK * 64;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 w = w1t[kk * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    if (TM == 1 && (r & 1)) h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], h2[t], 0, 0, 0);
                    else h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][kk][r], h[t], 0, 0, 0);
                }
        }
        const f32x4 bb = ld4(a.b1 + 16 * ht + 4 * lg);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (TM == 1) h[t] += h2[t];
            h[t] += bb;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[t][e] = gelu_bf(h[t][e]);
        }
#pragma unroll
        for (int o = 0; o < KK; ++o) {
            const f32x4 w = w2[((size_t)o * a.HT + ht) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    acc[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], h[t][r], acc[o][t], 0, 0, 0);
        }
    }

    // ---- 3. bias + residual, 16 B per lane -------------------------------------------------------
    // all loads first, then all stores: interleaved, the compiler must assume a store may alias the next load and waits out one
    // memory round trip per 64 bytes
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = m0 + t * 16 + l15;
        if (row >= a.M) continue;
        const float* xr = a.x + (size_t)row * CP + 4 * lg;
        float* orow = (a.out ? a.out : a.x) + (size_t)row * CP + 4 * lg;
        f32x4 res[KK];
#pragma unroll
        for (int o = 0; o < KK; ++o) { res[o] = ld4(xr + 16 * o); acc[o][t] += ld4(a.b2 + 16 * o + 4 * lg); }
#pragma unroll
        for (int o = 0; o < KK; ++o) st4(orow + 16 * o, res[o] + acc[o][t]);
    }
}

// ------------------------------------------------------------------------------------------------
// Block-cooperative variant for wide layers: the same register-resident chain per wave, but the NW waves of a
// workgroup share each hidden tile's 2*KK KiB of weight fragments through LDS.  The fragments are DMA'd
// global -> LDS with `global_load_lds_dwordx4` (lane-linear destination == fragment order, so no swizzle
// and conflict-free ds_read_b128), double-buffered: tile t+1 streams in while tile t is on the MFMA.
// L2 -> CU weight traffic drops NW-fold versus the wave-autonomous kernel, which is what wide layers need
// (at C = 384 one pass over fc1+fc2 is 4.7 MB).
// ------------------------------------------------------------------------------------------------
// Occupancy target: the loop has serial phases (DMA issue, GELU, barrier) that only a co-resident wave can cover, so the
// register budget is capped to fit 3-4 waves per SIMD where the tile sizes allow (s_memtime traces: with one wave per SIMD a
// hidden tile takes ~5600 cycles for 3072 cycles of MFMA work).
template <int CP, int TM> constexpr int mlp_min_waves() { return (CP * TM <= ESCX_MLP_OCC4) ? 4 : ((CP * TM <= 192) ? 3 : 1); }

// FC: compile the in-launch combine of the hidden split (a.tickets).  A SEPARATE instantiation on purpose: with that code in the body, hipcc
// allocates registers differently for the whole kernel and the plain (slabs + rows_combine_kernel) path of every hidden-split width gets
// 10-12 % slower (measured, profiles/r4_mlp_combine_ab.txt).
template <int CP, int TM, int NW, int ABL = 0, bool FC = false, bool SPLIT = false>      // ABL: timing-only ablation bits (never used by the product path)
__global__ __launch_bounds__(64 * NW, (mlp_min_waves<CP, TM>())) void mlp_fused_lds_kernel(MlpArgs a) {
    static_assert(!SPLIT || (TM == 1 && ABL == 0 && !FC), "the PatchSplit epilogue exists for the plain one-tile form only");
#ifdef ESCX_MLP_PRIO
    __builtin_amdgcn_s_setprio(ESCX_MLP_PRIO);      // tuning builds: static wave priority against co-running launches of the other batch part
#endif
    constexpr int KK = CP / 16;
    constexpr int CH = 2 * KK;                  // 1 KiB fragments per hidden tile: KK of fc1 then KK of fc2
    constexpr int PD = 3;                       // LDS -> register prefetch distance (fragments)
    __shared__ f32x4 wbuf[2][CH * 64];
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: keeps the DMA issue loop scalar
    const int HS = a.HS > 1 ? a.HS : 1;
    // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  The HS workgroups of one row block
    // sit 8 apart in the grid, so they share an XCD: the rows come from HBM once and the other HS-1 reads hit that L2.
    // (The last nrb % 8 row blocks are laid out plainly instead of padding the grid: a few padded workgroups would open a whole
    // extra dispatch round at one workgroup per CU.)
    int rb = blockIdx.x, hs = 0;
    if (HS > 1) {
        const int nrb = (a.M + 16 * TM * NW - 1) / (16 * TM * NW), full = nrb & ~7, i = blockIdx.x;
        if (i < full * HS) { const int grp = i / (8 * HS), r = i - grp * (8 * HS); rb = grp * 8 + (r & 7); hs = r >> 3; }
        else { const int j = i - full * HS; rb = full + j / HS; hs = j - (j / HS) * HS; }
    }
    const int ht0 = hs * (a.HT / HS), ht1 = ht0 + a.HT / HS;
    const int m0 = (rb * NW + wave) * (16 * TM);

    auto issue = [&](int ht, int buf) {
        const f32x4* src = a.wcf + (size_t)ht * CH * 64 + lane;
        for (int c = wave; c < CH; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&wbuf[buf][c * 64]), 16, 0, 0);
    };
    issue(ht0, ht0 & 1);

    f32x4 xf[TM][KK];
    float mean[TM], rstd[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = m0 + t * 16 + l15;
        const float* xr = a.x + (size_t)row * CP + 4 * lg;
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            xf[t][kk] = row < a.M ? ld4(xr + 16 * kk) : zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e) s += xf[t][kk][e];            // pad channels are exact zeros
        }
        s = sum_groups(s);
        mean[t] = s / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xf[t][kk][e] - mean[t]; v += d * d; }
        v = sum_groups(v) - (float)(CP - a.C) * mean[t] * mean[t];       // the zero pads each added mean^2
        rstd[t] = 1.0f / sqrtf(v / (float)a.C + a.eps);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const f32x4 g = ld4(a.gamma + 16 * kk + 4 * lg), b = ld4(a.beta + 16 * kk + 4 * lg);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xf[t][kk][e] = (xf[t][kk][e] - mean[t]) * rstd[t] * g[e] + b[e];    // gamma = beta = 0 in the pads -> 0
    }

    f32x4 acc[KK][TM];
#pragma unroll
    for (int o = 0; o < KK; ++o)
#pragma unroll
        for (int t = 0; t < TM; ++t) acc[o][t] = zero4();

    f32x4 bias_next = ld4(a.b1 + 16 * ht0 + 4 * lg);      // fc1 bias of the first tile; later tiles are fetched one stage ahead
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};
#define ESCX_TS(var) unsigned long long var = 0; if (ABL & 64) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    ESCX_TS(t_begin)
    for (int ht = ht0; ht < ht1; ++ht) {
        ESCX_TS(t0)
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                    // tile ht is in LDS for every wave; nobody still reads the other buffer
        }
        ESCX_TS(t1)
        // The next stage's DMA is not issued in one burst after the barrier (each 1 KiB global_load_lds costs 70-180 issue cycles during
        // which this wave feeds no MFMA, and the burst's landing slows the fc1 LDS reads) but one piece every DSTEP fragments of fc1.
        constexpr bool SPREAD = (ABL & 256) == 0;
        if (!SPREAD && ht + 1 < ht1 && !(ABL & 32)) issue(ht + 1, (ht + 1) & 1);
        // SPREAD: the last stage re-loads its own tile into the idle buffer (harmless) so that the loop body stays branch-free
        const f32x4* dsrc = (SPLIT && ht + 1 == ht1) ? a.sp_wf + lane                         // SPLIT: the stage after the last hidden tile is the first pair of split tiles
                                                     : a.wcf + (size_t)min(ht + 1, ht1 - 1) * CH * 64 + lane;
        f32x4* ddst = &wbuf[(ht + 1) & 1][0];
        ESCX_TS(t2)
        // The no-op pin makes the compiler wait for this tile's bias HERE, behind the vmcnt(0) above (free), instead of at its first
        // use after the fc1 MFMAs - where a vmcnt(0) would also wait out the DMA pieces issued in between (an L2 round trip per tile).
        f32x4 bb = bias_next;
        asm volatile("" : "+v"(bb));
        bias_next = ld4(a.b1 + 16 * min(ht + 1, ht1 - 1) + 4 * lg);
        const f32x4* wb = (ABL & 4) ? &wbuf[0][0] : &wbuf[ht & 1][lane];

        // fragment ring: the LDS read of fragment f + PD is in flight while fragment f feeds the MFMAs
        f32x4 ring[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) ring[i] = wb[i * 64];
        f32x4 h[TM], h2[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) { h[t] = bb; h2[t] = zero4(); }   // the fc1 bias rides in the accumulator (one VALU add less per tile)
#pragma unroll
        for (int f = 0; f < KK; ++f) {
            const f32x4 w = ring[f % PD];
            if (f + PD < CH) ring[f % PD] = wb[(f + PD) * 64];
            if constexpr (SPREAD && !(ABL & 32)) {             // one 1 KiB DMA every DSTEP fragments, all issued within the fc1 phase
                constexpr int NDMA = (CH + NW - 1) / NW, DSTEP = KK / NDMA > 0 ? KK / NDMA : 1;
                if (f % DSTEP == 0 && f / DSTEP < NDMA) {
                    int c = wave + (f / DSTEP) * NW;
                    if (CH % NW != 0) c = min(c, CH - 1);       // tail: a duplicate of the last piece (same bytes) keeps the loop body branch-free
                    __builtin_amdgcn_global_load_lds((const void*)(dsrc + c * 64), (__attribute__((address_space(3))) void*)(ddst + c * 64), 16, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    if (TM == 1 && (r & 1)) h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][f][r], h2[t], 0, 0, 0);
                    else h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], xf[t][f][r], h[t], 0, 0, 0);
                }
        }
        ESCX_TS(t3)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (TM == 1) h[t] += h2[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[t][e] = (ABL & 1) ? h[t][e] * 0.5f : gelu_bf(h[t][e]);
        }
        ESCX_TS(t4)
        // fc2: two output tiles per step so that consecutive MFMAs never hit the same accumulator
#pragma unroll
        for (int o = 0; o < KK; o += 2) {
            const int f = KK + o;
            const f32x4 w = ring[f % PD];
            if (f + PD < CH) ring[f % PD] = wb[(f + PD) * 64];
            f32x4 wn = zero4();
            if (o + 1 < KK) {
                wn = ring[(f + 1) % PD];
                if (f + 1 + PD < CH) ring[(f + 1) % PD] = wb[(f + 1 + PD) * 64];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    acc[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], h[t][r], acc[o][t], 0, 0, 0);
                    if (o + 1 < KK) acc[o + 1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[r], h[t][r], acc[o + 1][t], 0, 0, 0);
                }
        }
        ESCX_TS(t5)
        if (ABL & 64) { tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2; tr[3] += t4 - t3; tr[4] += t5 - t4; }
        // Pin the software pipeline: hipcc otherwise sinks every ds_read to just before its first use and
        // waits lgkmcnt(0) there, idling the matrix pipe for a full LDS round trip every 8 MFMAs.
        if (!(ABL & 64)) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // next tile's fc1 bias
#pragma unroll
            for (int i = 0; i < PD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int f = 0; f < CH; ++f) {
                if (f + PD < CH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if constexpr (SPREAD && !(ABL & 32)) {     // the DMA pieces stay where they are issued (unpinned, the scheduler sinks them to the end of the stage)
                    constexpr int NDMA = (CH + NW - 1) / NW, DSTEP = KK / NDMA > 0 ? KK / NDMA : 1;
                    if (f < KK && f % DSTEP == 0 && f / DSTEP < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM, 0);
            }
        }
    }
    if (ABL & 64) {
        ESCX_TS(t_end)
        tr[5] = t_end - t_begin;
        if (lane == 0 && a.trace) {
            unsigned long long* o = a.trace + (size_t)(blockIdx.x * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = tr[i];
            o[6] = t_begin; o[7] = t_end;
        }
    }
#undef ESCX_TS

    if (HS > 1 && (!FC || a.tickets == nullptr)) {       // raw fc2 partial sums; bias + residual are applied by rows_combine_kernel
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int row = m0 + t * 16 + l15;
            if (row >= a.M) continue;
            float* pr = a.partial + ((size_t)hs * a.M + row) * CP + 4 * lg;
#pragma unroll
            for (int o = 0; o < KK; ++o) st4(pr + 16 * o, acc[o][t]);
        }
        return;
    }
    if constexpr (FC) if (HS > 1) {
        // ---- combine by the last arriver (MI355X_MICROARCH.md, splitk-seam / publish-large rows; cdna_hip_programming.md Guideline 16 R1) ----
        // Publish: 16-byte sc1 (write-through) stores, so no release fence is needed; EVERY storing wave drains its stores, then ONE lane takes
        // the ticket with an agent-scope atomic.  The slab bytes are out of this XCD's L2 and in memory before the ticket is visible.
        const size_t slab = (size_t)a.M * CP;                   // floats per slab; HS * slab * 4 < 2^32 is checked by the launcher
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.partial, (unsigned)((size_t)HS * slab * sizeof(float)));
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int row = m0 + t * 16 + l15;
            if (row >= a.M) continue;
            const unsigned off = (unsigned)((((size_t)hs * a.M + row) * CP + 4 * lg) * sizeof(float));
#pragma unroll
            for (int o = 0; o < KK; ++o) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc[o][t]), rsrc, off + 64 * o, 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int ticket_s;
        __syncthreads();
        if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(a.tickets + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ticket_s != HS - 1) return;
        // Last arriver: every other slab of this row block is complete in memory.  ONE agent-scope acquire drops this CU's L1 (nobody on this
        // XCD has read these lines in this launch, and launch boundaries invalidate L2), then plain loads.  The counter goes back to zero for
        // the next launch (stream-ordered).
        if (threadIdx.x == 0) {
            __hip_atomic_store(a.tickets + rb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int row = m0 + t * 16 + l15;
            if (row >= a.M) continue;
            const float* xr = a.x + (size_t)row * CP + 4 * lg;
            f32x4 res[KK], v[KK];
#pragma unroll
            for (int o = 0; o < KK; ++o) res[o] = ld4(xr + 16 * o);
            for (int h2 = 0; h2 < HS; ++h2) {                   // slab-index order, NOT arrival order: bit-identical to rows_combine_kernel
                const float* pr = a.partial + ((size_t)h2 * a.M + row) * CP + 4 * lg;
#pragma unroll
                for (int o = 0; o < KK; ++o) {
                    const f32x4 p = (h2 == hs) ? acc[o][t] : ld4(pr + 16 * o);      // this workgroup's own slab is still in registers
                    v[o] = h2 == 0 ? p : v[o] + p;
                }
            }
            float* orow = a.x + (size_t)row * CP + 4 * lg;
#pragma unroll
            for (int o = 0; o < KK; ++o) { v[o] += ld4(a.b2 + 16 * o + 4 * lg); st4(orow + 16 * o, res[o] + v[o]); }
        }
        return;
    }
    if constexpr (SPLIT) {
        // ---- y = x + (mlp + b2) in registers (the plain epilogue's arithmetic), then PatchSplit: LayerNorm + linear + pixel-shuffled store ----
        const int row = m0 + l15;
        const bool live = row < a.M;
        const float* xr = a.x + (size_t)(live ? row : 0) * CP + 4 * lg;
        f32x4 y[KK];
#pragma unroll
        for (int o = 0; o < KK; ++o) { const f32x4 res = ld4(xr + 16 * o); acc[o][0] += ld4(a.b2 + 16 * o + 4 * lg); y[o] = live ? res + acc[o][0] : zero4(); }
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += y[kk][e];                    // pad channels are exact zeros
        sum = sum_groups(sum);
        const float smean = sum / (float)a.C;
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = y[kk][e] - smean; v += d * d; }
        v = sum_groups(v) - (float)(CP - a.C) * smean * smean;              // the zero pads each added mean^2
        const float srstd = 1.0f / sqrtf(v / (float)a.C + a.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f32x4 g = ld4(a.sp_gamma + 16 * kk + 4 * lg), bb = ld4(a.sp_beta + 16 * kk + 4 * lg);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[kk][e] = (y[kk][e] - smean) * srstd * g[e] + bb[e];       // gamma = beta = 0 in the pads
        }
        size_t ob0 = 0, ob1 = 0;
        if (live) {
            const int b = row / (a.sp_H * a.sp_W); const int r0 = row - b * a.sp_H * a.sp_W; const int h = r0 / a.sp_W, w = r0 - h * a.sp_W;
            ob0 = ((size_t)(b * 2 * a.sp_H + 2 * h) * a.sp_W + w) * a.sp_C2p;
            ob1 = ((size_t)(b * 2 * a.sp_H + 2 * h + 1) * a.sp_W + w) * a.sp_C2p;
        }
        const int n_st = a.sp_NT / 2;                                        // stages of two output tiles (CH = 2 KK fragments, the ring's stage size)
        for (int st = 0; st < n_st; ++st) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                                // split stage st is in LDS for every wave; nobody still reads the other buffer
            const int sbuf = (ht1 + st) & 1;
            if (st + 1 < n_st) {
                const f32x4* src = a.sp_wf + (size_t)(st + 1) * CH * 64 + lane;
                for (int c = wave; c < CH; c += NW)
                    __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&wbuf[sbuf ^ 1][c * 64]), 16, 0, 0);
            }
            const f32x4* wb = &wbuf[sbuf][lane];
            f32x4 o0 = zero4(), o1 = zero4();
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const f32x4 w0 = wb[kk * 64], w1 = wb[(KK + kk) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                               // one k-ordered chain per output tile (rowgemm_fused_kernel, K <= 192); the two tiles alternate
                    o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], y[kk][r], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], y[kk][r], o1, 0, 0, 0);
                }
            }
            if (live) {
                const int n0 = 16 * (2 * st) + 4 * lg, n1 = n0 + 16;
                const int s0 = n0 / a.sp_C2p, s1 = n1 / a.sp_C2p;
                st4(a.sp_out + (s0 ? ob1 : ob0) + (n0 - s0 * a.sp_C2p), o0);
                st4(a.sp_out + (s1 ? ob1 : ob0) + (n1 - s1 * a.sp_C2p), o1);
            }
        }
        return;
    }
    // all loads first, then all stores: interleaved, the compiler must assume a store may alias the next load and waits out one
    // memory round trip per 64 bytes
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = m0 + t * 16 + l15;
        if (row >= a.M) continue;
        const float* xr = a.x + (size_t)row * CP + 4 * lg;
        float* orow = (a.out ? a.out : a.x) + (size_t)row * CP + 4 * lg;
        f32x4 res[KK];
#pragma unroll
        for (int o = 0; o < KK; ++o) { res[o] = ld4(xr + 16 * o); acc[o][t] += ld4(a.b2 + 16 * o + 4 * lg); }
#pragma unroll
        for (int o = 0; o < KK; ++o) st4(orow + 16 * o, res[o] + acc[o][t]);
    }
}


// dst = src + (((P0 + P1) + ... + P_{n-1}) + bias) : fixed summation order, 16 B per lane, HBM-bound ((n + 2) * M * CP * 4 bytes).
// Second pass of the hidden-split MLP and of the head-group-split attention (dst may alias src).
__global__ __launch_bounds__(256) void rows_combine_kernel(float* dst, const float* src, const float* __restrict__ partial,
                                                           const float* __restrict__ bias, long long M, int CP, int n) {
    ESCX_SET_PRIO_SMALL();
    const long long n4 = M * CP / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % CP);
    f32x4 v = ld4(partial + i * 4);
    for (int h = 1; h < n; ++h) v += ld4(partial + ((size_t)h * M * CP) + i * 4);
    v += ld4(bias + c);
    st4(dst + i * 4, ld4(src + i * 4) + v);
}

}  // namespace escx
