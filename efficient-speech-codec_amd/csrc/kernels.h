// Non-GEMM kernels of the ESC hot path (gfx950): LayerNorm + gather, 4x4-window attention core,
// codebook search (distance + argmin), inverse-STFT overlap-add, layout converters.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"

namespace escx {

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows made of SEGS segments of Cp floats (C real channels each), optionally gathered.
//   MODE 0: identity rows               (norm2, PatchSplit.norm, PatchEmbed.norm)
//   MODE 1: window gather (SEGS == 1)   map[slot] = source token or -1; a -1 slot is a ZERO row placed
//                                       AFTER the norm (attention.py:135-143: pad follows norm1)
//   MODE 2: merge gather (SEGS == 2)    map[2*row+s] = source token or -1; a -1 segment is a zero row that
//                                       PARTICIPATES in the statistics (scale.py:106-112: pad precedes norm)
// 16 lanes per row, float4 per lane per step; two-pass mean / variance in registers-equivalent order.
// ------------------------------------------------------------------------------------------------
template <int SEGS, int MODE>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const int* __restrict__ map, int rows_per_clip, int src_rows_per_clip,
                                                      int total_rows, int C, int Cp, float eps) {
    const int sub = threadIdx.x & 15;
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int ngrp = (gridDim.x * 256) >> 4;
    const int V = Cp / 4;                       // float4 per segment
    for (int row = grp; row < total_rows; row += ngrp) {
        const int b = row / rows_per_clip, rr = row - b * rows_per_clip;
        const float* sp[SEGS];
        bool zero_out = false;
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
            int srow = (MODE == 0) ? rr : map[rr * SEGS + s];
            if (srow < 0) { sp[s] = nullptr; if (MODE == 1) zero_out = true; }
            else sp[s] = src + ((size_t)b * src_rows_per_clip + srow) * Cp;
        }
        float* dp = dst + (size_t)row * (SEGS * Cp);
        if (zero_out) {
            for (int v = sub; v < V; v += 16) st4(dp + 4 * v, zero4());
            continue;
        }
        // pass 1: mean over the real channels
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < SEGS; ++s)
            if (sp[s])
                for (int v = sub; v < V; v += 16) {
                    f32x4 x = ld4(sp[s] + 4 * v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (4 * v + e < C) sum += x[e];
                }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float mean = sum / (float)(SEGS * C);
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < SEGS; ++s)
            for (int v = sub; v < V; v += 16) {
                f32x4 x = sp[s] ? ld4(sp[s] + 4 * v) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) if (4 * v + e < C) { const float d = x[e] - mean; var += d * d; }
            }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) var += __shfl_xor(var, o, 16);
        const float rstd = 1.0f / sqrtf(var / (float)(SEGS * C) + eps);
#pragma unroll
        for (int s = 0; s < SEGS; ++s)
            for (int v = sub; v < V; v += 16) {
                f32x4 x = sp[s] ? ld4(sp[s] + 4 * v) : zero4();
                const f32x4 g = ld4(gamma + s * Cp + 4 * v), bb = ld4(beta + s * Cp + 4 * v);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (4 * v + e < C) ? (x[e] - mean) * rstd * g[e] + bb[e] : 0.f;
                st4(dp + s * Cp + 4 * v, y);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Window attention core (attention.py:222-241) for 4x4 windows: one wave per (window, head).
//   qkv : [B*nW*16][ldq]  columns: which*(nH*hdp) + h*hdp + d ; q is already scaled, pad dims are 0
//   bias: [nH][16][16]    relative-position bias gathered per head (attention.py:228-231)
//   out : [B*nW*16][ldo]  columns: h*hdp + d  (head concat, padded heads)
// S^T = K.Q^T on the MFMA (swapped so a lane holds one query row's scores for keys 4g..4g+3), softmax
// across the 4 lane groups with two shuffles, then O^T = V^T.P^T, again lane = (query, 4 consecutive d).
// STEPS = hdp/4 MFMA k-steps; lane (i, g) feeds dims STEPS*g .. STEPS*g+STEPS-1 (k-slot remap is free).
// ------------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(256) void window_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                               float* __restrict__ out, int total_pairs, int nH, int ldq, int ldo,
                                                               int nWh, int nWw, int shifted) {
    constexpr int HDP = 4 * STEPS;
    constexpr int DT = (HDP + 15) / 16;        // 16-wide output tiles per head
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int kOff = nH * HDP, vOff = 2 * nH * HDP;
    for (int pair = wave_global; pair < total_pairs; pair += nwaves) {
        const int win = pair / nH, h = pair - win * nH;
        const float* base = qkv + (size_t)win * 16 * ldq + h * HDP;
        const float* rowp = base + (size_t)i * ldq + STEPS * g;
        float kf[STEPS], qf[STEPS];
#pragma unroll
        for (int r = 0; r < STEPS; ++r) { qf[r] = rowp[r]; kf[r] = rowp[kOff + r]; }
        f32x4 s = zero4();
#pragma unroll
        for (int r = 0; r < STEPS; ++r) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[r], qf[r], s, 0, 0, 0);
        // lane (query i, group g) holds S[i][key 4g + r]
        s += ld4(bias + ((size_t)h * 16 + i) * 16 + 4 * g);
        if (shifted) {                          // attention.py:56-75 regions, evaluated on the fly
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
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = expf(s[r] - mx);
        float den = (p[0] + p[1]) + (p[2] + p[3]);
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
        const float inv = 1.0f / den;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] *= inv;
        // O^T[d][query] = sum_key V[key][d] * P[query][key]; k-slot (g, r) <-> key 4g + r
        float* orow = out + ((size_t)win * 16 + i) * ldo + h * HDP;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int d = t * 16 + i;
            f32x4 o = zero4();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float vv = (d < HDP) ? base[(size_t)(4 * g + r) * ldq + vOff + d] : 0.f;
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, p[r], o, 0, 0, 0);
            }
            if (t * 16 + 4 * g < HDP) st4(orow + t * 16 + 4 * g, o);
        }
        if (h == 0 && nH * HDP < ldo) {         // keep the K padding of the projection GEMM at exact zero
            for (int c = nH * HDP + g; c < ldo; c += 4) out[((size_t)win * 16 + i) * ldo + c] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Window attention for ANY window size (attention.py:215-244 with the mask of attention.py:56-75): the fallback of the unfused path for window_size != 4.  Every shipped
// configuration uses 4 x 4 windows (the MFMA kernels above / fused_attn.h); this form exists so that a checkpoint trained with another window size loads and runs.
// One wave per (window, head); a lane owns query rows lane, lane + 64, ... and walks the N = ws^2 keys three times (row maximum, denominator, P . V with the
// normalised probabilities - the reference's softmax-then-matmul order), sequential fp32 FMAs in dimension / key order.  q is pre-scaled by the Q-K-V GEMM's epilogue.
// bias: [head][N][N] gathered from the relative-position table at packing time.  Slots of padded positions hold zero rows (LayerNorm gather), exactly as the reference pads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void window_attention_any_kernel(const float* __restrict__ qkv, const float* __restrict__ bias, float* __restrict__ out, int nH, int hd,
                                                                  int hdp, int ldq, int ldo, int ws, int nWh, int nWw, int shift) {
    const int win = blockIdx.x, h = blockIdx.y, N = ws * ws;
    const int wloc = win % (nWh * nWw), wh = wloc / nWw, ww = wloc - wh * nWw;
    const int Hp = nWh * ws, Wp = nWw * ws;
    const float* base = qkv + (size_t)win * N * ldq + h * hdp;
    const int kOff = nH * hdp, vOff = 2 * nH * hdp;
    const float* bh = bias + (size_t)h * N * N;
    auto region = [&](int p, int P) { return p < P - ws ? 0 : (p < P - shift ? 1 : 2); };       // attention.py:59-64
    for (int i = threadIdx.x; i < N; i += 64) {
        const float* q = base + (size_t)i * ldq;
        const int labq = shift > 0 ? 3 * region(wh * ws + i / ws, Hp) + region(ww * ws + i % ws, Wp) : 0;
        auto score = [&](int j) {
            const float* k = base + (size_t)j * ldq + kOff;
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s = __builtin_fmaf(q[d], k[d], s);
            s += bh[(size_t)i * N + j];
            if (shift > 0) { const int labk = 3 * region(wh * ws + j / ws, Hp) + region(ww * ws + j % ws, Wp); s += (labk != labq) ? -100.0f : 0.0f; }
            return s;
        };
        float mx = -__builtin_inff();
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, score(j));
        float den = 0.f;
        for (int j = 0; j < N; ++j) den += expf(score(j) - mx);
        const float inv = 1.0f / den;
        float* orow = out + ((size_t)win * N + i) * ldo + h * hdp;
        for (int d0 = 0; d0 < hdp; d0 += 4) {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            for (int j = 0; j < N; ++j) {
                const float p = expf(score(j) - mx) * inv;
                const float* v = base + (size_t)j * ldq + vOff + d0;
                o0 = __builtin_fmaf(p, v[0], o0); o1 = __builtin_fmaf(p, v[1], o1); o2 = __builtin_fmaf(p, v[2], o2); o3 = __builtin_fmaf(p, v[3], o3);
            }
            orow[d0] = o0; orow[d0 + 1] = o1; orow[d0 + 2] = o2; orow[d0 + 3] = o3;     // dims hd .. hdp - 1: the V pad columns are zero
        }
        if (h == 0) for (int c = nH * hdp; c < ldo; ++c) out[((size_t)win * N + i) * ldo + c] = 0.f;     // K padding of the projection GEMM
    }
}

// ------------------------------------------------------------------------------------------------
// Codebook search (codebook.py:20-43): for 16 framed vectors of one product-VQ group per block,
//   z = sum of split-K partials (fixed order) ; zn = z / max(||z||, 1e-12)
//   dist[n] = (sum zn^2 - (2 zn).c_n) + ||c_n||^2  with c_n the pre-normalised code ; argmin, lowest index
//   wins ties, NaN wins over everything (torch.min semantics).
// The (2 zn).c^T products run on the MFMA with the CODEBOOK tile as the row operand so that a lane owns
// one vector and walks its codes in increasing index order; the 1024 x d codebook (<=128 KB) is read
// straight from L2 -- each fragment is used exactly once per block, so an LDS copy would add traffic.
// ------------------------------------------------------------------------------------------------
struct SearchArgs {
    const float* zpart; int splits; int M; int ldz;       // partials [splits][M][ldz]; group g at column g*dt
    const float* cbn; const float* c2; const float* cbraw; // [G][Ksz][dt], [G][Ksz], [G][Ksz][dt]
    int Ksz, d, Tq;
    long long* codes; long long bstride;                   // codes[b*bstride + g*Tq + t]
    float* loss; float loss_scale;                         // optional: loss[g*M + m] = scale * sum_j (cb[code][j]-z[j])^2 (reduced per clip, in fixed order, by loss_reduce_kernel)
    int l2norm;
};

__device__ __forceinline__ bool arg_better(float d1, int i1, float d2, int i2) {
    const bool n1 = d1 != d1, n2 = d2 != d2;
    if (n1 || n2) return n1 && (!n2 || i1 < i2);
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

template <int STEPS>
__global__ __launch_bounds__(256) void pvq_search_kernel(SearchArgs a) {
    ESCX_SET_PRIO_SMALL();
    constexpr int DT = 4 * STEPS;
    __shared__ float zs[16][DT + 1];
    __shared__ float zn2[16][DT];
    __shared__ float asum[16];
    __shared__ float bestd[4][16];
    __shared__ int besti[4][16];
    const int g = blockIdx.y;
    const int m0 = blockIdx.x * 16;
    const int tid = threadIdx.x;
    // 1) reduce split-K partials in a fixed order
    for (int e = tid; e < 16 * DT; e += 256) {
        const int r = e / DT, j = e - r * DT;
        const int m = m0 + r;
        // all slices are fetched before the first add (a load -> add -> load chain costs one memory round trip per slice); the
        // sum itself keeps the order 0, 1, 2, ... (pvq_down_splits caps the slice count at 16)
        float part[16];
        const bool live = m < a.M && j < a.d;
        const float* zp = a.zpart + (size_t)(live ? m : 0) * a.ldz + g * DT + (live ? j : 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) part[s] = (live && s < a.splits) ? zp[(size_t)s * a.M * a.ldz] : 0.f;
        float z = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) z += part[s];
        for (int s = 16; s < a.splits; ++s) z += live ? zp[(size_t)s * a.M * a.ldz] : 0.f;
        zs[r][j] = z;
    }
    __syncthreads();
    // 2) F.normalize(z) and sum(zn^2) (codebook.py:31-36); one thread per vector, sequential like a row reduction
    if (tid < 16) {
        float ss = 0.f;
        for (int j = 0; j < a.d; ++j) ss += zs[tid][j] * zs[tid][j];
        const float den = a.l2norm ? fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
        float s2 = 0.f;
        for (int j = 0; j < DT; ++j) {
            const float zn = (j < a.d) ? zs[tid][j] / den : 0.f;
            s2 += zn * zn;
            zn2[tid][j] = 2.0f * zn;
        }
        asum[tid] = s2;
    }
    __syncthreads();
    // 3) distances + running argmin
    const int lane = tid & 63, wave = tid >> 6;
    const int vi = lane & 15, lg = lane >> 4;
    float zf[STEPS];
#pragma unroll
    for (int r = 0; r < STEPS; ++r) zf[r] = zn2[vi][STEPS * lg + r];
    const float av = asum[vi];
    const float* cb = a.cbn + (size_t)g * a.Ksz * DT;
    const float* c2 = a.c2 + (size_t)g * a.Ksz;
    float bd = __builtin_inff();
    int bi = 0x7fffffff;
    bool have = false;
    const int per_wave = (a.Ksz + 3) / 4;
    const int cbeg = wave * per_wave, cend = min(a.Ksz, cbeg + per_wave);
    // Branch-free, one tile ahead: the codebook rows and the ||c||^2 of tile i+1 are in flight while tile i is on the MFMA.
    // (With the loads predicated per lane and c2[code] fetched inside the compare loop, every tile cost five serialised L2 round
    // trips: 40 us per launch for 5 us of work.)  Rows past the end are clamped, their codes are skipped by the range test below.
    auto load_tile = [&](int c0, float* cf, float* c2v) {
        const float* p = cb + (size_t)min(c0 + vi, a.Ksz - 1) * DT + STEPS * lg;
#pragma unroll
        for (int r = 0; r < STEPS; ++r) cf[r] = p[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) c2v[r] = c2[min(c0 + 4 * lg + r, a.Ksz - 1)];
    };
    float cf[STEPS], c2v[4], cfn[STEPS], c2n[4];
    if (cbeg < cend) load_tile(cbeg, cf, c2v);
    for (int c0 = cbeg; c0 < cend; c0 += 16) {
        load_tile(min(c0 + 16, cend - 1), cfn, c2n);            // the last iteration re-reads its own tile (harmless)
        f32x4 dot = zero4();
#pragma unroll
        for (int r = 0; r < STEPS; ++r) dot = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[r], zf[r], dot, 0, 0, 0);
        // lane (vector vi, group lg) holds dot for codes c0 + 4*lg + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int code = c0 + 4 * lg + r;
            if (code < cend) {
                const float dist = (av - dot[r]) + c2v[r];
                if (!have || arg_better(dist, code, bd, bi)) { bd = dist; bi = code; have = true; }
            }
        }
#pragma unroll
        for (int r = 0; r < STEPS; ++r) cf[r] = cfn[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) c2v[r] = c2n[r];
    }
    if (!have) { bd = __builtin_inff(); bi = 0x7fffffff; }
    // across the 4 lane groups of the wave
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float od = __shfl_xor(bd, o);
        const int oi = __shfl_xor(bi, o);
        if (arg_better(od, oi, bd, bi)) { bd = od; bi = oi; }
    }
    if (lg == 0) { bestd[wave][vi] = bd; besti[wave][vi] = bi; }
    __syncthreads();
    if (tid < 16) {
        float d0 = bestd[0][tid]; int i0 = besti[0][tid];
        for (int w = 1; w < 4; ++w) if (arg_better(bestd[w][tid], besti[w][tid], d0, i0)) { d0 = bestd[w][tid]; i0 = besti[w][tid]; }
        const int m = m0 + tid;
        if (m < a.M) {
            const int b = m / a.Tq, t = m - b * a.Tq;
            a.codes[(size_t)b * a.bstride + (size_t)g * a.Tq + t] = (long long)i0;
            if (a.loss) {       // eval-mode commitment loss: mse(z_q, z_e).mean([1,2]) / groups (codebook.py:72-73)
                const float* q = a.cbraw + ((size_t)g * a.Ksz + i0) * DT;
                float e = 0.f;
                for (int j = 0; j < a.d; ++j) { const float df = q[j] - zs[tid][j]; e += df * df; }
                a.loss[(size_t)g * a.M + m] = e * a.loss_scale;       // no atomics: the per-clip sum must be run-to-run deterministic
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PVQ framing + residual + down-projection, split-K slices (quantization.py:74-91, 388-410; csrvq.py:16-18), round 4:
//     zpart[z][(b, t)][n] = sum over k in slice z of W_down[n][k] * (enc - dec)[(b, h, ov*t + o)][c],   k = (o, h, c)
// the specialised form of gemm_kernel<64, BN, BK, ResidualGatherA, EpiPartial>: a wave owns 16 framed vectors and one K slice; per 16-wide k chunk the
// (o, h, c) decomposition is wave-uniform scalar arithmetic (the engine's loader did two multiply-high divisions per 16-byte operand fetch), the operand
// and the NT weight fragments come straight from global memory one chunk ahead.  Same contraction order as the engine (chunks ascending, MFMA r = 0..3,
// one chain per output tile), same slice boundaries (k_per_z from pvq_down_splits): bit-identical partial sums.
// ------------------------------------------------------------------------------------------------
#ifdef ESCX_EXPERIMENTAL
struct PvqDownArgs { const float* enc; const float* dec; const float* W; float* zpart; int M, Tq, Hq, Wd, Cp, ov, Kp, Np, k_per_z; };

template <int NT>
__global__ __launch_bounds__(256) void pvq_down_kernel(PvqDownArgs a) {
    ESCX_SET_PRIO_SMALL();
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = (blockIdx.x * 4 + wave) * 16 + l15;
    const bool live = m < a.M;
    const int b = live ? m / a.Tq : 0, t = live ? m - b * a.Tq : 0;
    const size_t vecbase = ((size_t)b * a.Hq * a.Wd + (size_t)a.ov * t) * a.Cp + 4 * lg;
    const int kbeg = blockIdx.y * a.k_per_z, kend = min(a.Kp, kbeg + a.k_per_z);
    const float* wrow = a.W + (size_t)l15 * a.Kp + 4 * lg;
    auto load = [&](int k0, f32x4& af, f32x4 (&wf)[NT]) {
        const int oh = k0 / a.Cp, cc = k0 - oh * a.Cp, o = oh / a.Hq, h = oh - o * a.Hq;       // wave-uniform: a chunk never straddles a (o, h) row (Cp % 16 == 0)
        const size_t idx = vecbase + (size_t)(h * a.Wd + o) * a.Cp + cc;
        af = live ? ld4(a.enc + idx) : zero4();
        if (a.dec && live) af -= ld4(a.dec + idx);
#pragma unroll
        for (int n = 0; n < NT; ++n) wf[n] = ld4(wrow + (size_t)(16 * n) * a.Kp + k0);
    };
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = zero4();
    f32x4 af, wf[NT], afn, wfn[NT];
    if (kbeg < kend) load(kbeg, af, wf);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        load(min(k0 + 16, kend - 16), afn, wfn);                // the last chunk re-reads itself (harmless)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][r], af[r], acc[n], 0, 0, 0);
        af = afn;
#pragma unroll
        for (int n = 0; n < NT; ++n) wf[n] = wfn[n];
    }
    if (live) {
        float* zr = a.zpart + ((size_t)blockIdx.y * a.M + m) * a.Np + 4 * lg;
#pragma unroll
        for (int n = 0; n < NT; ++n) st4(zr + 16 * n, acc[n]);
    }
}

#endif  // ESCX_EXPERIMENTAL

// ------------------------------------------------------------------------------------------------
// PVQ de-quantisation + up-projection + un-frame + residual add (quantization.py:93-108, 124-136, 412-432; csrvq.py:19-21), round 4:
//     out[(b, h, ov*t + o)][c] = dec[...] + sum_k W_up[n = (o, h, c)][k] * cb_g(k)[code[b, g(k), t]][k - g(k)*dt]
// The GEMM engine ran this as a generic tile kernel (CodeGatherA loader + EpiPvqAdd epilogue): with K = 32..96 it issued 7-10 VALU instructions
// per MFMA on index arithmetic (three multiply-high divisions per 16-byte store, a dependent code load per operand fetch) - 16-22 % MFMA-busy,
// 30-43 us alone on the GPU for 20 us of HBM traffic, and 3.2x that next to the other batch part.  Here a wave gathers its 16 vectors' codebook
// rows ONCE into the MFMA operand, then walks output tiles: per tile KC weight fragments straight from L2, 4*KC MFMAs, and a store whose
// address is a wave-uniform tile offset plus a per-lane row base.  The contraction order is the engine's (16-wide k chunks ascending, within
// a chunk MFMA r = 0..3 with k = chunk + 4*slot + r, one accumulator chain, then + dec): bit-identical results.
// ------------------------------------------------------------------------------------------------
struct PvqUpArgs {
    const long long* codes; long long bstride; const float* cb; const float* W; const float* dec; float* out;
    int G, Ksz, dt, Tq, M, Hq, Wd, Cp, ov, Kup, NT, nt_per_wg;
};

template <int KC>
__global__ __launch_bounds__(256) void pvq_up_kernel(PvqUpArgs a) {
    ESCX_SET_PRIO_SMALL();
    constexpr int UNR = 4;                      // output tiles in flight per wave
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 16 + l15;
    const bool live = m < a.M;
    const int b = live ? m / a.Tq : 0, t = live ? m - b * a.Tq : 0;
    const size_t base = ((size_t)b * a.Hq * a.Wd + (size_t)a.ov * t) * a.Cp + 4 * lg;
    f32x4 zf[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = 16 * c + 4 * lg, g = k / a.dt;
        zf[c] = zero4();
        if (live && g < a.G) {
            long long code = a.codes[(size_t)b * a.bstride + (size_t)g * a.Tq + t];
            code = code < 0 ? 0 : (code >= a.Ksz ? a.Ksz - 1 : code);       // a corrupt index must not read outside the codebook (F.embedding would raise)
            zf[c] = ld4(a.cb + ((size_t)g * a.Ksz + (size_t)code) * a.dt + (k - g * a.dt));
        }
    }
    const int nt_lo = blockIdx.y * a.nt_per_wg, nt_hi = min(a.NT, nt_lo + a.nt_per_wg);
    const float* wrow = a.W + (size_t)l15 * a.Kup + 4 * lg;
    for (int nt0 = nt_lo + wave * UNR; nt0 < nt_hi; nt0 += 4 * UNR) {
        f32x4 wf[UNR][KC], dv[UNR];
        size_t idx[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int nt = min(nt0 + u, nt_hi - 1);                            // ragged tail: a duplicate tile, not stored
#pragma unroll
            for (int c = 0; c < KC; ++c) wf[u][c] = ld4(wrow + (size_t)(16 * nt) * a.Kup + 16 * c);
            const int n0 = 16 * nt, oh = n0 / a.Cp, c0 = n0 - oh * a.Cp, o = oh / a.Hq, h = oh - o * a.Hq;      // wave-uniform
            idx[u] = base + (size_t)(h * a.Wd + o) * a.Cp + c0;
            dv[u] = (live && a.dec) ? ld4(a.dec + idx[u]) : zero4();
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f32x4 acc = zero4();
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][c][r], zf[c][r], acc, 0, 0, 0);
            if (live && nt0 + u < nt_hi) {
                if (a.dec) acc += dv[u];
                st4(a.out + idx[u], acc);
            }
        }
    }
}

// cm_loss[b] = sum over (stream slot, group, frame) of the per-vector terms written by pvq_search_kernel, in a fixed order:
// lane l of the clip's wave adds terms l, l+64, ... in increasing index order, then a fixed butterfly joins the 64 lanes.
// terms: [n_slots][G][M] with M = B*Tq rows laid out (b, t).
__global__ __launch_bounds__(64) void loss_reduce_kernel(const float* __restrict__ terms, int n_slots, int G, int M, int Tq, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int per = n_slots * G * Tq;
    float acc = 0.f;
    for (int i = lane; i < per; i += 64) {
        const int sg = i / Tq, t = i - sg * Tq;
        acc += terms[(size_t)sg * M + (size_t)b * Tq + t];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = acc;
}

// ------------------------------------------------------------------------------------------------
// Inverse STFT tail: overlap-add of windowed inverse-DFT frames and division by the window envelope
// (torch.istft semantics: center=True trims n_fft/2, length = hop*(T-1)).
//   frames: [B*T][ldf] holds w[j] * irfft(frame)[left + j], j in [0, win)
// ------------------------------------------------------------------------------------------------
__global__ void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win2, float* __restrict__ wave,
                                 int B, int T, int ldf, int win, int hop, int left, int half, int out_len) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * out_len) return;
    const int b = idx / out_len, s = idx - b * out_len;
    const int p = s + half - left;               // position relative to the start of frame 0's window support
    int t_hi = p / hop; if (t_hi > T - 1) t_hi = T - 1;
    float acc = 0.f, env = 0.f;
    for (int t = t_hi; t >= 0; --t) {
        const int j = p - t * hop;
        if (j >= win) break;
        acc += frames[((size_t)b * T + t) * ldf + j];
        env += win2[j];
    }
    wave[idx] = acc / env;
}

// ------------------------------------------------------------------------------------------------
// Border pixels of the composed de-embedding.  The 3x3 convolution zero-pads the FINE (pixel-shuffled) map, so
// outputs on the first/last fine row/column must drop the neighbours that fall outside; they use a variant of the
// composed 7x7 weights (escx_api.cpp: compose_deembed).  One wave per border coarse pixel, K split over lanes.
//   wv: [16 variants][NO][49][C] unpadded, bv: [16][NO]; variant = 4*eh + ew, e = 1 first, 2 last, 3 both.
// ------------------------------------------------------------------------------------------------
template <int NO>
__global__ __launch_bounds__(256) void deembed_border_kernel(const float* __restrict__ x, const float* __restrict__ wv,
                                                             const float* __restrict__ bv, float* __restrict__ out, int B, int H, int W,
                                                             int C, int Cp, int pf, int pt, int in_dim, int Fp) {
    const int per_clip = 2 * W + 2 * (H > 2 ? H - 2 : 0);          // rows 0 and H-1 in full, then the two side columns
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * per_clip) return;
    const int b = wave / per_clip; int i = wave - b * per_clip;
    int h, w;
    if (i < W) { h = 0; w = i; }
    else if (i < 2 * W) { h = H - 1; w = i - W; }
    else { i -= 2 * W; h = 1 + (i >> 1); w = (i & 1) ? W - 1 : 0; }
    if (H == 1 && i >= W && i < 2 * W) return;                       // the single row was already handled
    if (W == 1 && i >= 2 * W && (i & 1)) return;
    const int eh = (h == 0 ? 1 : 0) | (h == H - 1 ? 2 : 0), ew = (w == 0 ? 1 : 0) | (w == W - 1 ? 2 : 0);
    const int K = 49 * C;
    const float* wvar = wv + (size_t)(4 * eh + ew) * NO * K;
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.f;
    for (int k = lane; k < K; k += 64) {
        const int tap = k / C, ci = k - tap * C;
        const int hh = h + tap / 7 - 3, ww = w + tap % 7 - 3;
        if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
        const float xv = x[((size_t)(b * H + hh) * W + ww) * Cp + ci];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = fmaf(wvar[(size_t)o * K + k], xv, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) acc[o] += __shfl_xor(acc[o], sft);
    }
    if (lane == 0) {
        const int Q = pf * pt;
        for (int o = 0; o < NO; ++o) {
            const int co = o / Q, q = o - co * Q, s1 = q / pt, s2 = q - s1 * pt;
            out[((size_t)(b * (pt * W) + pt * w + s2) * in_dim + co) * Fp + pf * h + s1] = acc[o] + bv[(4 * eh + ew) * NO + o];
        }
    }
}

// layout converters between reference (unpadded) rows and internal padded rows
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int C, int Cp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * Cp) return;
    const long long r = idx / Cp; const int c = (int)(idx - r * Cp);
    dst[idx] = c < C ? src[r * C + c] : 0.f;
}
__global__ void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int C, int Cp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const long long r = idx / C; const int c = (int)(idx - r * C);
    dst[idx] = src[r * Cp + c];
}
// Wire format of the codes: every code is log2(codebook_size) = 10 bits, so 6 x 3 x 50 codes/s x 10 b = 9 kbps
// exactly (base.py:70).  4 codes -> 5 bytes, little-endian bit order; n is padded to a multiple of 4 with zero codes.
__global__ void codes_pack10_kernel(const long long* __restrict__ in, unsigned char* __restrict__ out, long long n) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    unsigned long long v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long i = q * 4 + j; v |= (unsigned long long)((i < n ? in[i] : 0) & 1023) << (10 * j); }
#pragma unroll
    for (int j = 0; j < 5; ++j) out[q * 5 + j] = (unsigned char)(v >> (8 * j));
}
__global__ void codes_unpack10_kernel(const unsigned char* __restrict__ in, long long* __restrict__ out, long long n) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    unsigned long long v = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) v |= (unsigned long long)in[q * 5 + j] << (8 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long i = q * 4 + j; if (i < n) out[i] = (long long)((v >> (10 * j)) & 1023); }
}
// Calibration kernel for the HBM counters (tools/pmc_calib.py): a copy with the fused kernels' access pattern - 16 lanes per row, one
// 16-byte access per lane and step, rows of Cp floats (192 B at Cp = 48) - over buffers far larger than the 256 MiB Infinity Cache.
__global__ __launch_bounds__(256) void test_copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int Cp) {
    const int sub = threadIdx.x & 15;
    const long long grp = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4, ngrp = ((long long)gridDim.x * 256) >> 4;
    for (long long r = grp; r < rows; r += ngrp)
        for (int v = sub; v < Cp / 4; v += 16) st4(dst + r * Cp + 4 * v, ld4(src + r * Cp + 4 * v));
}
__global__ void test_math_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int which) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = which == 0 ? gelu_bf(x[i]) : (which == 1 ? erf_bf(x[i]) : (which == 2 ? exp_fast(x[i]) : gelu_erf(x[i])));
}
__global__ void codes_narrow_kernel(const long long* __restrict__ in, short* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (short)in[i];
}
__global__ void codes_widen_kernel(const short* __restrict__ in, long long* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (long long)in[i];
}

}  // namespace escx
