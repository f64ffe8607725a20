// Halo-tiled composed de-embedding for gfx950 (scale.py:73-81 folded into one 7x7 convolution, see escx_api.cpp):
//
//   out[b, pt*w + s2, co, pf*h + s1] = bias[n] + sum_{tap (dh,dw) in 7x7} sum_ci  Wc[n][tap][ci] * x[b, h+dh-3, w+dw-3, ci],   n = (co, s1, s2)
//
// The implicit-GEMM form (gemm_kernel + ConvA) gathers every input pixel 49 times from L2: 6.5 GB per 36-clip step for a
// 133 MB input.  Here a workgroup owns a TH x TW tile of coarse pixels, brings the tile plus a 3-pixel halo into LDS ONCE
// (zero outside the map), and every tap reads its operand from there with a shifted base address:
//   * x operand of the MFMA: lane (pixel l&15 of a 16-pixel row segment, k-slot group g = l>>4) reads 16 B = channels
//     16kk + 4g .. +3 of pixel (row + dh, col + dw + l&15) - one ds_read_b128; the pixel stride is CP + 4 dwords, which
//     spreads 16 consecutive pixels over distinct bank quads (conflict-free);
//   * weights: host-packed MFMA fragments [tap][kk][lane][4] (row operand = the 16 padded outputs n), streamed through a
//     double-buffered LDS ring one tap row (7 taps) per stage with global_load_lds_dwordx4, shared by the 8 waves;
//   * a wave owns one tile row = TM = 2 segments of 16 pixels that share each weight fragment; D leaves lane
//     (pixel, outputs 4g..4g+3), stored straight into the frame-major spectrum.
// Pixels on the first/last row/column of the map need other weights (the 3x3 zero-pads the FINE map); the border kernel
// overwrites them afterwards, exactly as with the implicit-GEMM form.
// Algorithmic traffic: read x once (+ halo), write the spectrum once.  Bound: MFMA (2 * 49 * CP * 16 FLOP per pixel).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gemm_engine.h"
#include "split_terms.h"

namespace escx {

struct DeembedArgs {
    const float* x;             // [B][H][W][CP] tokens (h = frequency patch, w = time patch)
    const f32x4* wf;            // [49][KK][64] weight fragments (interior variant)
    const float* bias;          // [16]
    float* out;                 // frame-major spectrum [B][pt*W][in_dim][Fp]
    int B, H, W, pf, pt, in_dim, Fp, n_out;
};

template <int CP>
__global__ __launch_bounds__(512) void deembed7_kernel(DeembedArgs a) {
    ESCX_SET_PRIO_SMALL();
    constexpr int KK = CP / 16, TH = 8, TW = 32, HH = TH + 6, HW = TW + 6, PS = CP + 4;     // PS: pixel stride in dwords
    constexpr int NW = 8, NP = 7 * KK, NPW = (NP + NW - 1) / NW;
    __shared__ float xs[HH * HW * PS];
    __shared__ f32x4 wr[2][NP * 64];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntw = (a.W + TW - 1) / TW, nth = (a.H + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tw = bid % ntw; bid /= ntw;
    const int th = bid % nth; const int b = bid / nth;
    const int h0 = th * TH, w0 = tw * TW;

    auto issue_row = [&](int dh, int buf) {     // one tap row of weights: NP pieces of 1 KiB
        const f32x4* src = a.wf + (size_t)dh * NP * 64 + lane;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            int c = wave + i * NW;
            if (NP % NW != 0) c = min(c, NP - 1);
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&wr[buf][c * 64]), 16, 0, 0);
        }
    };
    issue_row(0, 0);

    // ---- tile + halo -> LDS (zeros outside the map) ------------------------------------------------
    // Tap row dh reads halo rows dh .. dh + TH - 1, so only the first TH rows are needed up front; row TH + dh is fetched during
    // step dh (at most one 16-byte vector per thread) and written to LDS before the next barrier.
    const float* xb = a.x + (size_t)b * a.H * a.W * CP;
    auto halo_vec = [&](int v) -> f32x4 {       // vector v of the halo tile: pixel v / (4 KK), channels 4 * (v % (4 KK)) ..
        const int pix = v / (KK * 4), q = v - pix * (KK * 4);
        const int ph = pix / HW, pw = pix - ph * HW;
        const int gh = h0 - 3 + ph, gw = w0 - 3 + pw;
        return (gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) ? ld4(xb + ((size_t)gh * a.W + gw) * CP + 4 * q) : zero4();
    };
    auto halo_put = [&](int v, f32x4 val) {
        const int pix = v / (KK * 4), q = v - pix * (KK * 4);
        *reinterpret_cast<f32x4*>(&xs[pix * PS + 4 * q]) = val;
    };
    constexpr int ROWV = HW * KK * 4;           // vectors per halo row
    static_assert(ROWV <= 512, "one vector per thread per extra halo row");
    for (int v = tid; v < TH * ROWV; v += 512) halo_put(v, halo_vec(v));

    // ---- 49 taps ------------------------------------------------------------------------------------
    const bool seg1 = w0 + 16 < a.W;            // the last column tile of a 16-aligned-but-not-32-aligned map has one segment only
    f32x4 acc[2] = {zero4(), zero4()};
    const float* xrow = &xs[(wave * HW + l15) * PS + 4 * lg];          // this wave's tile row, pixel l15 of segment 0, tap (0, 0)
    auto tap_row = [&](auto two_segments, int dh) {
        constexpr bool TWO = decltype(two_segments)::value;
        constexpr int NS = 7 * KK;              // (dw, kk) steps; the LDS reads of step s + 1 are in flight while step s feeds the MFMAs
        const f32x4* wb = &wr[dh & 1][lane];
        const float* xr = xrow + dh * HW * PS;
        f32x4 w[2], x0[2], x1[2];
        auto fetch = [&](int st, int slot) {
            const int dw = st / KK, kk = st - dw * KK;
            w[slot] = wb[st * 64];
            x0[slot] = *reinterpret_cast<const f32x4*>(xr + dw * PS + 16 * kk);
            if (TWO) x1[slot] = *reinterpret_cast<const f32x4*>(xr + (dw + 16) * PS + 16 * kk);
            __builtin_amdgcn_sched_group_barrier(0x100, TWO ? 3 : 2, 0);
        };
        fetch(0, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + 1 < NS) fetch(st + 1, (st + 1) & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[st & 1][r], x0[st & 1][r], acc[0], 0, 0, 0);
                if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[st & 1][r], x1[st & 1][r], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TWO ? 8 : 4, 0);
        }
    };
    for (int dh = 0; dh < 7; ++dh) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // tap row dh and halo row TH + dh - 1 are in LDS; nobody still reads the other weight buffer
        issue_row(min(dh + 1, 6), (dh + 1) & 1);
        const bool extra = dh < 6 && tid < ROWV;
        f32x4 nv = zero4();
        if (extra) nv = halo_vec((TH + dh) * ROWV + tid);
        if (seg1) tap_row(std::true_type{}, dh); else tap_row(std::false_type{}, dh);
        if (extra) halo_put((TH + dh) * ROWV + tid, nv);
    }

    // ---- bias + store: lane (pixel, outputs 4lg .. 4lg+3) -----------------------------------------
    const int h = h0 + wave;
    if (h >= a.H) return;
    const int Q = a.pf * a.pt;
    const f32x4 bv = ld4(a.bias + 4 * lg);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int w = w0 + 16 * t + l15;
        if (w >= a.W) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = 4 * lg + r;
            if (nn >= a.n_out) continue;
            const int co = nn / Q, q = nn - co * Q, s1 = q / a.pt, s2 = q - s1 * a.pt;
            a.out[((size_t)(b * (a.pt * a.W) + a.pt * w + s2) * a.in_dim + co) * a.Fp + a.pf * h + s1] = acc[t][r] + bv[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Two-term fp16 form (round 5, split_terms.h; the spectrum it writes feeds the ISTFT only - never a code).  The contraction is flattened over all 49 taps:
// k = tap * CP + c, 32 per MFMA step (ceil(49 CP / 32) steps: no per-tap padding of the 48-channel map to 64).  A lane's eight k-slots 32 s + 8 g .. + 7 lie inside ONE tap
// (CP % 8 == 0), so every lane group walks its own (dh, dw, c) through the halo tile.  The tile + halo sits in LDS ONCE as two fp16 planes (split when it is loaded,
// 112 B per pixel: 16 consecutive pixels start in 16 distinct 16-byte granules of the 256-byte LDS row); the weights - two fp16 terms per step, scaled by the power of
// two of max |w| (deembed7_x2_pack_kernel, from the fp32 fragment stream) - stream through a double-buffered LDS ring in chunks of 8 steps (16 KiB).
// Per step and wave: 2 weight + 4 operand fragment reads, 6 MFMAs (terms (0,1), (1,0), (0,0) x two 16-pixel segments).  LDS: 116 + 32 KiB.
// ------------------------------------------------------------------------------------------------
constexpr int de2_steps(int CP) { return (49 * CP + 31) / 32; }
inline size_t deembed7_x2_bytes(int CP) { return (size_t)de2_steps(CP) * 2 * 1024 + 32; }      // + trailer: bits of max |w|, {2^-k, 2^k}

// one thread per (step, lane): lane (n, g) of step s holds the two terms of W[n][tap][c .. c + 7], k = 32 s + 8 g = tap CP + c (zero behind the last tap)
__global__ __launch_bounds__(256) void deembed7_x2_pack_kernel(const f32x4* __restrict__ wf, bf16x8* __restrict__ out, int CP, int nsteps) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nsteps * 64) return;
    const int step = idx >> 6, lane = idx & 63, n = lane & 15, g = lane >> 4, KK = CP / 16;
    bf16x8* tail = out + (size_t)nsteps * 2 * 64;
    const float sc = x2_scale(*reinterpret_cast<const unsigned*>(tail));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * step + 8 * g + e, tap = k / CP, c = k - tap * CP;
        v[e] = tap < 49 ? wf[(size_t)(tap * KK + c / 16) * 64 + 16 * ((c % 16) / 4) + n][c % 4] : 0.f;
    }
    bf16x8 t[2];
    split_terms<2>(v, t, sc);
    out[(size_t)(step * 2 + 0) * 64 + lane] = t[0]; out[(size_t)(step * 2 + 1) * 64 + lane] = t[1];
    if (idx == 0) { float* o = reinterpret_cast<float*>(tail + 1); o[0] = 1.0f / sc; o[1] = sc; o[2] = 0.f; o[3] = 0.f; }
}

template <int CP>
__global__ __launch_bounds__(512) void deembed7_x2_kernel(DeembedArgs a, const bf16x8* __restrict__ wf2) {
    static_assert(CP == 48, "one wrap of the channel index per 32-deep step");
    ESCX_SET_PRIO_SMALL();
    constexpr int TH = 8, TW = 32, HH = TH + 6, HW = TW + 6, PS = CP + 8;        // PS: halfs per pixel
    constexpr int NSTEP = de2_steps(CP), CH = 8, NCH = (NSTEP + CH - 1) / CH, NW = 8, Q8 = CP / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char de2_lds[];
    _Float16* xh0 = reinterpret_cast<_Float16*>(de2_lds);                       // [HH][HW][PS] high terms
    _Float16* xh1 = xh0 + HH * HW * PS;                                          // ... low terms
    bf16x8* wr = reinterpret_cast<bf16x8*>(xh1 + HH * HW * PS);                  // [2][CH * 2 * 64]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntw = (a.W + TW - 1) / TW, nth = (a.H + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tw = bid % ntw; bid /= ntw;
    const int th = bid % nth; const int b = bid / nth;
    const int h0 = th * TH, w0 = tw * TW;

    auto issue_chunk = [&](int ch, int buf) {
        const int pieces = min(CH, NSTEP - ch * CH) * 2;
        const bf16x8* src = wf2 + (size_t)ch * CH * 2 * 64 + lane;
        for (int c = wave; c < pieces; c += NW)
            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(&wr[(buf * CH * 2 + c) * 64]), 16, 0, 0);
    };
    issue_chunk(0, 0);
    const f32x4 scv = *reinterpret_cast<const f32x4*>(wf2 + (size_t)NSTEP * 2 * 64 + 1);       // {2^-k, 2^k, 0, 0}

    // ---- tile + halo -> two fp16 planes (zero outside the map) ----
    // Range rule (split_terms.h): the decoder tokens reach this kernel un-normalised (scale.py:73-81 has no norm in front of de_proj1), so no bound is known when the image is
    // packed: the workgroup scales ITS tile by the power of two that brings the tile's own max |x| into [2^13, 2^14) before the split and the accumulators back afterwards
    // (exact; a pixel that lies in two tiles' halos may be split under two scales - either way to 2^-22 of the tile's largest value).
    const float* xb = a.x + (size_t)b * a.H * a.W * CP;
    constexpr int NV = (HH * HW * Q8 + 511) / 512;
    float xr[NV][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 512 * i;
        const int pix = v / Q8, q = v - pix * Q8;
        const int ph = pix / HW, pw = pix - ph * HW;
        const int gh = h0 - 3 + ph, gw = w0 - 3 + pw;
        if (v < HH * HW * Q8 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) {
            const float* p = xb + ((size_t)gh * a.W + gw) * CP + 8 * q;
            const f32x4 u0 = ld4(p), u1 = ld4(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xr[i][e] = u0[e]; xr[i][4 + e] = u1[e]; amax = fmaxf(amax, fmaxf(fabsf(u0[e]), fabsf(u1[e]))); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[i][e] = 0.f;
        }
    }
    __shared__ float de2_red[NW];
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) de2_red[wave] = amax;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NW; ++i) amax = fmaxf(amax, de2_red[i]);
    const float act_sc = act_pow2_scale(amax), act_inv = 1.0f / act_sc;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 512 * i;
        if (v < HH * HW * Q8) {
            const int pix = v / Q8, q = v - pix * Q8;
            bf16x8 t[2];
            split_terms<2>(xr[i], t, act_sc);
            *reinterpret_cast<bf16x8*>(xh0 + pix * PS + 8 * q) = t[0];
            *reinterpret_cast<bf16x8*>(xh1 + pix * PS + 8 * q) = t[1];
        }
    }

    const bool seg1 = w0 + 16 < a.W;
    f32x4 acc0 = zero4(), acc1 = zero4();
    int c = 8 * lg, dw = 0, dh = 0;             // this lane group's position in the flattened contraction: k = (7 dh + dw) CP + c
    for (int ch = 0; ch < NCH; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // chunk ch (and, the first time, the halo planes) are in LDS; nobody still reads the other ring slot
        if (ch + 1 < NCH) issue_chunk(ch + 1, (ch + 1) & 1);
        const bf16x8* wb = &wr[((ch & 1) * CH * 2) * 64 + lane];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (ch * CH + s >= NSTEP) break;
            const int dhc = dh > 6 ? 6 : dh;    // behind the last tap the weights are zero: any valid address
            const int xoff = ((wave + dhc) * HW + dw + l15) * PS + c;
            const bf16x8 wt0 = wb[(s * 2) * 64], wt1 = wb[(s * 2 + 1) * 64];
            const bf16x8 x00 = *reinterpret_cast<const bf16x8*>(xh0 + xoff), x01 = *reinterpret_cast<const bf16x8*>(xh1 + xoff);
            acc0 = mma_x<2>(wt0, x01, acc0); acc0 = mma_x<2>(wt1, x00, acc0); acc0 = mma_x<2>(wt0, x00, acc0);
            if (seg1) {
                const bf16x8 x10 = *reinterpret_cast<const bf16x8*>(xh0 + xoff + 16 * PS), x11 = *reinterpret_cast<const bf16x8*>(xh1 + xoff + 16 * PS);
                acc1 = mma_x<2>(wt0, x11, acc1); acc1 = mma_x<2>(wt1, x10, acc1); acc1 = mma_x<2>(wt0, x10, acc1);
            }
            c += 32;
            if (c >= CP) { c -= CP; if (++dw == 7) { dw = 0; ++dh; } }
        }
    }

    // ---- scale back, bias, store: lane (pixel, outputs 4lg .. 4lg+3) - as deembed7_kernel ----
    const int h = h0 + wave;
    if (h >= a.H) return;
    const int Q = a.pf * a.pt;
    const f32x4 bv = ld4(a.bias + 4 * lg);
    const f32x4 accs[2] = {acc0 * (scv[0] * act_inv), acc1 * (scv[0] * act_inv)};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int w = w0 + 16 * t + l15;
        if (w >= a.W) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = 4 * lg + r;
            if (nn >= a.n_out) continue;
            const int co = nn / Q, q = nn - co * Q, s1 = q / a.pt, s2 = q - s1 * a.pt;
            a.out[((size_t)(b * (a.pt * a.W) + a.pt * w + s2) * a.in_dim + co) * a.Fp + a.pf * h + s1] = accs[t][r] + bv[r];
        }
    }
}

}  // namespace escx
