// One product-VQ stream of the cross-scale quantiser in ONE launch (round 5):
//   frame + residual (enc - dec)  ->  down-projection  ->  normalise  ->  codebook search (argmin)  ->  de-quantise  ->  up-projection
//   -> un-frame + add                                    (quantization.py:74-136, 388-432; codebook.py:20-55; csrvq.py:15-21, 50-60)
//
// Until round 4 this was three launches per stream and batch part (split-K down-projection GEMM -> 33 MB of partial sums in HBM ->
// pvq_search_kernel, which re-reduced them -> pvq_up_kernel): 18 + 6 launches per step and part, 0.85 ms alone on the GPU and 1.87 ms of
// stream time in the product's two-stream execution, where every short launch queues behind the other part's resident workgroups.
//
// Geometry.  A workgroup owns 16 framed vectors (one MFMA row tile) and has 16 waves:
//   P1  wave z IS split-K slice z of the down-projection (the slices of the engine form: k_per_z from pvq_down_splits, so the slicing depends
//       on the layer geometry only).  Its partial tile goes to LDS, not to HBM.
//   P2  the slices are added per element in slice order 0, 1, 2, ... (the order of pvq_search_kernel's reduce: bit-identical z), then one
//       thread per (vector, group) normalises exactly as before (sequential fmaf chain, IEEE sqrt and division).
//   P3  wave w searches codes [64 w, 64 w + 64) of every group: (2 zn) . c^T on the MFMA with the normalised codebook tile as the row operand,
//       dist = (sum zn^2 - dot) + ||c||^2 in that order (codebook.py:35-39), running argmin in increasing code order, lowest index wins a tie,
//       NaN semantics of torch.min (arg_better is a strict total order, so the result does not depend on how the code range is cut);
//       wavefront shuffles across the four lane groups, then the 16 waves' candidates meet in LDS.
//   P4  the 16 code vectors are gathered once per wave (raw codebook rows, codebook.py:52-53) and the waves walk the output tiles of the
//       up-projection: weights straight from L2, + dec, un-framed store.  Contraction order of pvq_up_kernel: bit-identical output.
// Why 16 vectors per workgroup and not 64: at 18 clips per batch part a stream has 2700 vectors = 169 row tiles; one tile per workgroup
// puts them on 169 CUs in one dispatch round, each moving ~0.3-0.65 MB of rows and as much of weights.  64 vectors per workgroup would re-read
// the projection matrices 4x less often from L2 but run on 43 CUs (2.6 MB per CU for the 9600-token map: a longer critical path).
// Nothing is shared between the slices of P1 (disjoint k ranges), so there is nothing to stage through LDS there; the codebook tile of P3 is
// used once per workgroup (16 vectors) and is read straight from L2 as before.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"
#include "kernels.h"

namespace escx {

struct PvqFusedArgs {
    const float* enc; const float* dec;             // residual = enc - dec (dec may be null: first stream)
    const float* wd;                                // down-projection in MFMA fragment order [k chunk][n tile][lane][4] (Quant::wdf), k = (o, h, c)
    const float* cbn; const float* c2; const float* cbraw;      // [G][Ksz][dt] normalised, [G][Ksz] squared norms, [G][Ksz][dt] raw
    const float* wup;                               // [Kq][Kup = Np] up-projection (used when there is no table)
    const float* tab; const float* gq;              // de-quantisation table [(h, ov * code + o)][Cp] and float4 -> group map (Quant::tab / gq), or nullptr
    float* out;                                     // nullptr: codes only (last requested stream, csrvq.py:151); may alias dec
    long long* codes; long long bstride;            // codes[b * bstride + g * Tq + t]
    float* loss; float loss_scale;                  // optional per-vector commitment terms, loss[g * M + m]
    int M, Tq, Hq, Wd, Cp, ov, Kq, k_per_z, splits;
    int G, Ksz, d, l2norm;
    unsigned long long* trace;                      // tuning builds (-DESCX_PVQ_TRACE): 8 s_memtime stamps per workgroup
};

constexpr int PVQF_WAVES = 16;
constexpr int PVQF_GMAX = 4;
template <int NT> constexpr int pvqf_lds_floats() { return PVQF_WAVES * 16 * (16 * NT + 4); }

// DEC: a.dec != nullptr (a template parameter, not a run-time test: with branches around the loads hipcc falls back to `s_waitcnt vmcnt(0)`
// before every use and the register rings below degenerate to one memory round trip per chunk - measured: 44 us for the 20 chunks of the
// 9600-token map).  All load sections are therefore branch-free: addresses are clamped into valid memory, unused values are discarded by selects.
template <int NT, int STEPS, bool DEC>
__global__ __launch_bounds__(64 * PVQF_WAVES) void pvq_fused_kernel(PvqFusedArgs a) {
#pragma clang fp contract(off)
    ESCX_SET_PRIO_SMALL();
    constexpr int NP = 16 * NT, NPS = NP + 4, DT = 4 * STEPS, KC = NT;
    extern __shared__ __attribute__((aligned(16))) float pvqf_part[];            // [16 slices][16 rows][NPS]
    __shared__ float zs[16][NP + 1];
    __shared__ float zn2[PVQF_GMAX][16][DT];
    __shared__ float asum[PVQF_GMAX][16];
    __shared__ float bestd[PVQF_WAVES][PVQF_GMAX][16];
    __shared__ int besti[PVQF_WAVES][PVQF_GMAX][16];
    __shared__ int code_s[PVQF_GMAX][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 16;
    const int m = m0 + l15;
    const bool live = m < a.M;
    const int b = live ? m / a.Tq : 0, t = live ? m - b * a.Tq : 0;                 // rows past M alias vector 0: valid memory, their values are discarded
    const size_t vecbase = ((size_t)b * a.Hq * a.Wd + (size_t)a.ov * t) * a.Cp + 4 * lg;
#ifdef ESCX_PVQ_TRACE
#define PVQ_TS(i) if (a.trace && tid == 0) { __builtin_amdgcn_sched_barrier(0); a.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PVQ_TS(i)
#endif
    PVQ_TS(0)

    // ---- P1: split-K slice `wave` of the down-projection (pvq_down_kernel's contraction: chunks of 16 ascending, MFMA r = 0..3, one chain per tile) ----
    // Bound by memory LATENCY, not by bytes: the rows come from HBM (~2 us per dependent round trip under load), so a wave keeps PF chunks (rows of
    // enc and dec + the NT weight fragments of each) in flight in a register ring; the MFMA chain itself stays strictly k-ascending.
    {
        constexpr int PF = NT >= 6 ? 2 : (NT >= 4 ? 3 : (NT == 3 ? 4 : (NT == 2 ? 5 : 6)));
        const int kbeg = wave * a.k_per_z, kend = min(a.Kq, kbeg + a.k_per_z);
        if (wave < a.splits && kbeg < kend) {                  // wave-uniform
            const int nch = (kend - kbeg) >> 4;
            const float* wfrag = a.wd + (size_t)lane * 4;                                       // fragment (chunk, tile) = 1 KiB contiguous: whole cache lines per fetch
            f32x4 er[PF], dr[PF], wr[PF][NT];
            auto issue = [&](int ci, f32x4& e, f32x4& dd, f32x4 (&wf)[NT]) {
                const int k0 = kbeg + 16 * min(ci, nch - 1);                                   // past the end: re-read the last chunk (never consumed)
                const int oh = k0 / a.Cp, cc = k0 - oh * a.Cp, o = oh / a.Hq, h = oh - o * a.Hq;       // wave-uniform: a chunk never straddles a (o, h) row (Cp % 16 == 0)
                const size_t idx = vecbase + (size_t)(h * a.Wd + o) * a.Cp + cc;
                e = ld4(a.enc + idx);
                if constexpr (DEC) dd = ld4(a.dec + idx);
#pragma unroll
                for (int n = 0; n < NT; ++n) wf[n] = ld4(wfrag + ((size_t)(k0 >> 4) * NT + n) * 256);
            };
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = zero4();
            auto consume = [&](const f32x4& e, const f32x4& dd, const f32x4 (&wf)[NT]) {
                f32x4 af = e;
                if constexpr (DEC) af -= dd;
                if (!live) af = zero4();                        // a select, not a branch
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][r], af[r], acc[n], 0, 0, 0);
            };
#pragma unroll
            for (int j = 0; j < PF; ++j) issue(j, er[j], dr[j], wr[j]);
            const int rounds = nch / PF, tail = nch - rounds * PF;
            for (int rd = 0; rd < rounds; ++rd) {               // straight-line body: the compiler's vmcnt waits are exact (PF - 1 chunks stay in flight)
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    consume(er[j], dr[j], wr[j]);
                    issue((rd + 1) * PF + j, er[j], dr[j], wr[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < PF - 1; ++j)
                if (j < tail) consume(er[j], dr[j], wr[j]);     // wave-uniform
            float* pr = pvqf_part + ((size_t)wave * 16 + l15) * NPS + 4 * lg;
#pragma unroll
            for (int n = 0; n < NT; ++n) st4(pr + 16 * n, acc[n]);
        }
    }
    __syncthreads();
    PVQ_TS(1)

    // ---- P2: slices added in slice order (z = ((0 + p0) + p1) + ..., as pvq_search_kernel), then F.normalize and sum(zn^2) per (vector, group) ----
    for (int e = tid; e < 16 * NP; e += 64 * PVQF_WAVES) {
        const int r = e / NP, n = e - r * NP;
        float z = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) z += (s < a.splits) ? pvqf_part[((size_t)s * 16 + r) * NPS + n] : 0.f;
        zs[r][n] = z;
    }
    __syncthreads();
    if (tid < 16 * a.G) {
        const int g = tid >> 4, vi = tid & 15;
        float zr[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) zr[j] = zs[vi][g * DT + j];               // all LDS reads first (g * DT + j < NP), then the sequential chains
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < DT; ++j) ss = (j < a.d) ? __builtin_fmaf(zr[j], zr[j], ss) : ss;
        const float den = a.l2norm ? fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const float zn = (j < a.d) ? zr[j] / den : 0.f;
            s2 = __builtin_fmaf(zn, zn, s2);
            zn2[g][vi][j] = 2.0f * zn;
        }
        asum[g][vi] = s2;
    }
    __syncthreads();
    PVQ_TS(2)

    // ---- P3: distances + running argmin; wave w takes codes [w * per_wave, (w + 1) * per_wave) of every group ----
    // TB code tiles (all four of a 1024-entry codebook's 64 codes per wave) are fetched together: one L2 round trip per group.  The running update is
    // branch-free: a lane visits its codes in increasing order, so "candidate is better" (arg_better with i1 > i2) is  !(d >= best) && best == best  -
    // d < best, or d is NaN while best is not; a NaN best is never replaced (the lowest index wins, torch.min).
    {
        constexpr int TB = 4;
        const int per_wave = ((a.Ksz + PVQF_WAVES - 1) / PVQF_WAVES + 15) & ~15;
        const int cbeg = wave * per_wave, cend = min(a.Ksz, cbeg + per_wave);
        for (int g = 0; g < a.G; ++g) {
            float zf[STEPS];
#pragma unroll
            for (int r = 0; r < STEPS; ++r) zf[r] = zn2[g][l15][STEPS * lg + r];
            const float av = asum[g][l15];
            const float* cb = a.cbn + (size_t)g * a.Ksz * DT;
            const float* c2 = a.c2 + (size_t)g * a.Ksz;
            float bd = __builtin_inff();
            int bi = 0x7fffffff;
            bool have = false;
            for (int cb0 = cbeg; cb0 < cend; cb0 += 16 * TB) {
                float cf[TB][STEPS], c2v[TB][4];
#pragma unroll
                for (int u = 0; u < TB; ++u) {                          // rows past the end are clamped, their codes are skipped by the range test below
                    const int c0 = cb0 + 16 * u;
                    const float* p = cb + (size_t)min(c0 + l15, a.Ksz - 1) * DT + STEPS * lg;
#pragma unroll
                    for (int r = 0; r < STEPS; ++r) cf[u][r] = p[r];
#pragma unroll
                    for (int r = 0; r < 4; ++r) c2v[u][r] = c2[min(c0 + 4 * lg + r, a.Ksz - 1)];
                }
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    const int c0 = cb0 + 16 * u;
                    f32x4 dot = zero4();
#pragma unroll
                    for (int r = 0; r < STEPS; ++r) dot = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[u][r], zf[r], dot, 0, 0, 0);
                    // lane (vector l15, group lg) holds dot for codes c0 + 4*lg + r
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int code = c0 + 4 * lg + r;
                        const float dist = (av - dot[r]) + c2v[u][r];
                        const bool in = code < cend;
                        const bool take = in & (!have | (!(dist >= bd) & (bd == bd)));
                        bd = take ? dist : bd; bi = take ? code : bi; have = have | in;
                    }
                }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {            // across the 4 lane groups of the wave (lanes without a code carry (+inf, INT_MAX))
                const float od = __shfl_xor(bd, o);
                const int oi = __shfl_xor(bi, o);
                const bool n1 = od != od, n2 = bd != bd;
                const bool better = (n1 | n2) ? (n1 & (!n2 | (oi < bi))) : ((od < bd) | ((od == bd) & (oi < bi)));      // arg_better, without branches
                bd = better ? od : bd; bi = better ? oi : bi;
            }
            if (lg == 0) { bestd[wave][g][l15] = bd; besti[wave][g][l15] = bi; }
        }
    }
    __syncthreads();
    PVQ_TS(3)
    if (tid < 16 * a.G) {
        const int g = tid >> 4, vi = tid & 15;
        float wd_[PVQF_WAVES]; int wi_[PVQF_WAVES];
#pragma unroll
        for (int w = 0; w < PVQF_WAVES; ++w) { wd_[w] = bestd[w][g][vi]; wi_[w] = besti[w][g][vi]; }
        float d0 = wd_[0]; int i0 = wi_[0];
#pragma unroll
        for (int w = 1; w < PVQF_WAVES; ++w) {
            const bool n1 = wd_[w] != wd_[w], n2 = d0 != d0;
            const bool better = (n1 | n2) ? (n1 & (!n2 | (wi_[w] < i0))) : ((wd_[w] < d0) | ((wd_[w] == d0) & (wi_[w] < i0)));
            d0 = better ? wd_[w] : d0; i0 = better ? wi_[w] : i0;
        }
        code_s[g][vi] = i0;
        const int mm = m0 + vi;
        if (mm < a.M) {
            const int bb = mm / a.Tq, tt = mm - bb * a.Tq;
            a.codes[(size_t)bb * a.bstride + (size_t)g * a.Tq + tt] = (long long)i0;
            if (a.loss) {       // eval-mode commitment loss: mse(z_q, z_e).mean([1,2]) / groups (codebook.py:72-73)
                const float* q = a.cbraw + ((size_t)g * a.Ksz + i0) * DT;
                float e = 0.f;
                for (int j = 0; j < a.d; ++j) { const float df = q[j] - zs[vi][g * DT + j]; e = __builtin_fmaf(df, df, e); }
                a.loss[(size_t)g * a.M + mm] = e * a.loss_scale;       // no atomics: the per-clip sum must be run-to-run deterministic
            }
        }
    }
    if (!a.out) return;
    __syncthreads();
    PVQ_TS(4)

    // ---- P4, table form: out = dec + tab[(h, ov * code_g + o)][c] - the up-projection of every code was evaluated once, by pvq_up_kernel, when the
    // table was built (bit-identical); at run time the phase is pure data movement, UNR tiles in flight per wave, loads branch-free ----
    if (a.tab) {
        constexpr int UNR = 8;
        const int NTo = a.Kq / 16;
        const int tabw = a.ov * a.Ksz;                                                         // frames of the table's pseudo clip
        int cs[PVQF_GMAX];
#pragma unroll
        for (int g = 0; g < PVQF_GMAX; ++g) cs[g] = code_s[g < a.G ? g : 0][l15];
        for (int nt0 = wave * UNR; nt0 < NTo; nt0 += PVQF_WAVES * UNR) {
            f32x4 tv[UNR], dv[UNR];
            size_t idx[UNR];
            int gsel[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int nt = min(nt0 + u, NTo - 1);                              // ragged tail: a duplicate tile, not stored
                const int n0 = 16 * nt, oh = n0 / a.Cp, c0 = n0 - oh * a.Cp, o = oh / a.Hq, h = oh - o * a.Hq;      // wave-uniform
                idx[u] = vecbase + (size_t)(h * a.Wd + o) * a.Cp + c0;
                gsel[u] = (int)a.gq[(n0 >> 2) + lg];
                const int gg = gsel[u] < 0 ? 0 : gsel[u];
                const int code = gg == 0 ? cs[0] : (gg == 1 ? cs[1] : (gg == 2 ? cs[2] : cs[3]));
                tv[u] = ld4(a.tab + ((size_t)h * tabw + (size_t)a.ov * code + o) * a.Cp + c0 + 4 * lg);       // padding float4s read a valid row and are zeroed below
                if constexpr (DEC) dv[u] = ld4(a.dec + idx[u]);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                f32x4 v = gsel[u] >= 0 ? tv[u] : zero4();
                if constexpr (DEC) v += dv[u];
                if (live && nt0 + u < NTo) st4(a.out + idx[u], v);
            }
        }
        PVQ_TS(5)
        return;
    }
    // ---- P4, MFMA form (no table for this geometry, or ESCX_PVQ_TABLE=0): de-quantise (raw codebook rows) + up-projection + un-frame + add ----
    {
        constexpr int UNR = KC >= 6 ? 2 : (KC >= 4 ? 3 : (KC == 3 ? 4 : 6));       // output tiles in flight per wave within the 128-register budget of a 1024-thread workgroup
        f32x4 zf[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = 16 * c + 4 * lg, g = k / DT;
            zf[c] = zero4();
            if (live && g < a.G) zf[c] = ld4(a.cbraw + ((size_t)g * a.Ksz + (size_t)code_s[g][l15]) * DT + (k - g * DT));
        }
        const int NTo = a.Kq / 16;
        const float* wrow = a.wup + (size_t)l15 * NP + 4 * lg;
        for (int nt0 = wave * UNR; nt0 < NTo; nt0 += PVQF_WAVES * UNR) {
            f32x4 wf[UNR][KC], dv[UNR];
            size_t idx[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int nt = min(nt0 + u, NTo - 1);                              // ragged tail: a duplicate tile, not stored
#pragma unroll
                for (int c = 0; c < KC; ++c) wf[u][c] = ld4(wrow + (size_t)(16 * nt) * NP + 16 * c);
                const int n0 = 16 * nt, oh = n0 / a.Cp, c0 = n0 - oh * a.Cp, o = oh / a.Hq, h = oh - o * a.Hq;      // wave-uniform
                idx[u] = vecbase + (size_t)(h * a.Wd + o) * a.Cp + c0;
                if constexpr (DEC) dv[u] = ld4(a.dec + idx[u]);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                f32x4 acc = zero4();
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][c][r], zf[c][r], acc, 0, 0, 0);
                if constexpr (DEC) acc += dv[u];
                if (live && nt0 + u < NTo) st4(a.out + idx[u], acc);
            }
        }
    }
    PVQ_TS(5)
#undef PVQ_TS
}

// De-quantise + up-projection + un-frame + add of one stream as a table-row add (the decode path: codes given).  One thread per float4 of the
// output map, coalesced along the channels; reads dec + one table row segment, writes out: HBM-bound.
struct PvqTabAddArgs {
    const long long* codes; long long bstride; const float* tab; const float* gq; const float* dec; float* out;
    int G, Ksz, Tq, Hq, Wd, Cp, ov; long long n4;
};
__global__ __launch_bounds__(256) void pvq_tab_add_kernel(PvqTabAddArgs a) {
    ESCX_SET_PRIO_SMALL();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n4) return;
    const int c4 = a.Cp >> 2;
    const long long row = i / c4; const int c = (int)(i - row * c4) * 4;                       // row = (b, h, w)
    const int w = (int)(row % a.Wd); const long long bh = row / a.Wd; const int h = (int)(bh % a.Hq); const int b = (int)(bh / a.Hq);
    const int t = w / a.ov, o = w - t * a.ov;
    const int g = (int)a.gq[(((o * a.Hq + h) * a.Cp) + c) >> 2];
    f32x4 v = zero4();
    if (g >= 0) {
        long long code = a.codes[(size_t)b * a.bstride + (size_t)g * a.Tq + t];
        code = code < 0 ? 0 : (code >= a.Ksz ? a.Ksz - 1 : code);       // a corrupt index must not read outside the table (F.embedding would raise)
        v = ld4(a.tab + ((size_t)h * a.ov * a.Ksz + (size_t)a.ov * code + o) * a.Cp + c);
    }
    if (a.dec) v += ld4(a.dec + i * 4);
    st4(a.out + i * 4, v);
}

}  // namespace escx
