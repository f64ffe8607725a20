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

}  // namespace escx
