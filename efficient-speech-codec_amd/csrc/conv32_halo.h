// 32 -> 32-channel convolutions of the multi-resolution discriminators (discriminator.py:139-177: five Conv2d per band, (3, 9) and (3, 3) taps, stride (1, 1) or
// (1, 2)) as a correlation over an INPUT TILE HELD IN LDS (round 4).  The implicit-GEMM engine gathers every input element once per tap - 27 times through
// L1 / L2 for a (3, 9) kernel - and with 32 output channels a staged activation tile feeds only two MFMA column tiles: those launches ran at 80-87 TFLOP/s,
// bound by the gather traffic, not by the matrix pipe.  Here a workgroup owns 4 x 32 output positions, loads the (4 + T0 - 1) x (31 s1 + T1) input positions
// they touch ONCE (zero-filled outside the map), and its four waves (one output frame each, two 16-row tiles) walk the taps over that image; the 32 x 32
// weight slice of a tap is staged through a ring of three LDS slots (the slice two taps ahead is in flight), one barrier per tap.
//
// Same arithmetic as the engine, in the same order: for every output element the products are added tap-major / channel-minor (k = (t0 T1 + t1) 32 + c) on
// v_mfma_f32_16x16x4_f32, zero-padded taps contribute exact zeros - results are bit-identical to gemm_kernel<.., ConvSU / ConvTS / ConvTSP, ..> on the same
// operands (checked: tests/test_disc.py against the oracle and the fixtures, ESCX_CONV32_HALO=0 = the engine form).
//
// MEASURED (round 4, adversarial step at 36 clips): bit-identical hashes of all 108 feature maps, 324 parameter gradients and the waveform gradient
// (tools/disc_ab.py); alone the 32 -> 32 layers go 86 -> 90 TFLOP/s forward and 86 -> 89-95 dX, but the step gets SLOWER, 323.3 -> 335.5 ms (two runs each,
// alternating): 70 KB of LDS and 23 us workgroups next to the other stream's launches - the pattern of section 8.2 of DESIGN.md again.  Opt-in
// (ESCX_CONV32_HALO=1); the engine form stays the default.
//
// One kernel serves forward and dX: input position of output (o0, o1) and tap (t0, t1) is (o0 + off0 + d0 t0, o1 s1 + off1 + d1 t1); forward: d = +1,
// off = -pad; dX of a stride-1 layer: d = -1, off = +pad; dX of one residue class of a strided layer (ConvTSP): s1 = 1, d = -1, off = (c0, c1) over the
// class's compact tap set.  The epilogues are the engine's (EpiConvOut, EpiAccumView, EpiAccumPhase): rows are numbered m = (b O0 + o0) O1 + o1 as there.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "disc_kernels.h"
#include "gemm_bf16.h"

namespace escx {

struct Halo32 {
    TView x; int T0, T1, s1, d0, d1, off0, off1, O0, O1, B;
    int ok;                     // host: geometry is covered (32 channels in and out, unit stride along the frames)
    int ok16;                   // ... with 16 (padded) output channels: dX of the 2 -> 32 first layers (bf16 form only)
};

constexpr int H32_TO0 = 4, H32_TO1 = 32, H32_P = 36;           // output tile; LDS floats per input position (32 channels + 4: fragment reads of consecutive positions spread over the banks)

#ifdef ESCX_EXPERIMENTAL       // the fp32 form of the tile kernel: tagged builds only (tune_env.h)
template <class Epi>
__global__ __launch_bounds__(256) void conv32_halo_kernel(Halo32 g, const float* __restrict__ W, int Kp, int tiles0, int tiles1, Epi ep) {
    extern __shared__ __attribute__((aligned(16))) float h32_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int NF = H32_TO0 + g.T0 - 1, HW = (H32_TO1 - 1) * g.s1 + g.T1;
    float* Hs = h32_lds;                                        // [NF][HW][P]
    float* Ws = h32_lds + (size_t)NF * HW * H32_P;              // [3][32][P]
    int bid = blockIdx.x;
    const int t1i = bid % tiles1; bid /= tiles1;
    const int t0i = bid % tiles0; const int bi = bid / tiles0;
    const int o0b = t0i * H32_TO0, o1b = t1i * H32_TO1;
    const int lo0 = g.d0 < 0 ? -(g.T0 - 1) : 0, lo1 = g.d1 < 0 ? -(g.T1 - 1) : 0;      // most negative tap displacement
    const int in0b = o0b + g.off0 + lo0, in1b = o1b * g.s1 + g.off1 + lo1;             // input position of LDS image element (0, 0)
    // input image -> LDS, zero outside the map
    const int total = NF * HW * 8;
    const float* xb = g.x.p + (size_t)bi * g.x.D0 * g.x.P1 * g.x.Cp;
    for (int i = tid; i < total; i += 256) {
        const int c4 = i & 7, pos = i >> 3;
        const int h0 = pos / HW, h1 = pos - h0 * HW;
        const int i0 = in0b + h0, i1 = in1b + h1;
        const bool ok = (unsigned)i0 < (unsigned)g.x.D0 && (unsigned)i1 < (unsigned)g.x.D1;
        const f32x4 v = ok ? ld4(xb + ((size_t)i0 * g.x.P1 + i1) * g.x.Cp + 4 * c4) : zero4();
        st4(Hs + (size_t)pos * H32_P + 4 * c4, v);
    }
    const int wn = tid >> 3, wc = 4 * (tid & 7);                // weight staging role: row n, channels wc .. wc + 3 of the tap's slice
    const int NT = g.T0 * g.T1;
    const float* wsrc = W + (size_t)wn * Kp + wc;
    float* wdst = Ws + wn * H32_P + wc;
    // weight slices: ring of three LDS slots, the slice of tap + 2 is in flight (a register) while tap is on the MFMA: two taps of compute per L2 round trip
    st4(wdst, ld4(wsrc));
    if (NT > 1) st4(wdst + 32 * H32_P, ld4(wsrc + 32));
    f32x4 wreg = NT > 2 ? ld4(wsrc + 64) : zero4();
    __syncthreads();

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = zero4();
    int t0 = 0, t1 = 0, slot = 0;
    for (int tap = 0; tap < NT; ++tap) {
        f32x4 wnew = zero4();
        if (tap + 3 < NT) wnew = ld4(wsrc + (tap + 3) * 32);
        const float* hrow = Hs + ((size_t)(wave + g.d0 * t0 - lo0) * HW + (g.d1 * t1 - lo1)) * H32_P + 4 * lg;
        const float* wrow = Ws + slot * 32 * H32_P + l15 * H32_P + 4 * lg;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            f32x4 af[2], wf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) af[b] = ld4(hrow + (size_t)((16 * b + l15) * g.s1) * H32_P + 16 * kc);
#pragma unroll
            for (int a = 0; a < 2; ++a) wf[a] = ld4(wrow + 16 * a * H32_P + 16 * kc);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a][r], af[b][r], acc[a][b], 0, 0, 0);
        }
        // the slice of tap + 2 goes into the slot tap - 1 used: every wave left that slot before the previous barrier
        const int wslot = slot == 0 ? 2 : slot - 1;
        if (tap + 2 < NT) st4(wdst + wslot * 32 * H32_P, wreg);
        if (tap + 1 < NT) __syncthreads();
        wreg = wnew;
        slot = slot == 2 ? 0 : slot + 1;
        if (++t1 == g.T1) { t1 = 0; ++t0; }
    }
    const int o0 = o0b + wave;
    if (o0 >= g.O0) return;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int o1 = o1b + 16 * b + l15;
        if (o1 >= g.O1) continue;
        const int m = (bi * g.O0 + o0) * g.O1 + o1;
#pragma unroll
        for (int a = 0; a < 2; ++a) ep.store(m, 16 * a + 4 * lg, acc[a][b], 0);
    }
}
#endif  // ESCX_EXPERIMENTAL

// The same tile on the bf16 MFMA (the discriminator's opt-in bf16 precision, gemm_bf16.h): the input image and the weight slices are rounded to bf16 when they are
// staged (80-byte LDS rows: one ds_read_b128 = the 8 channels of a lane's k slots), a tap is ONE 32-deep MFMA step per accumulator tile; fp32 accumulation in tap order.
constexpr int H32_P16 = 40;
template <int TN, class Epi>                // TN: 16-wide tiles of output channels (2: the 32 -> 32 layers; 1: dX of the 2 -> 32 first layers, 16 padded input channels)
__global__ __launch_bounds__(256) void conv32_halo_bf16_kernel(Halo32 g, const float* __restrict__ W, int Kp, int tiles0, int tiles1, Epi ep) {
    extern __shared__ __attribute__((aligned(16))) __bf16 h16_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int NF = H32_TO0 + g.T0 - 1, HW = (H32_TO1 - 1) * g.s1 + g.T1;
    __bf16* Hs = h16_lds;                                       // [NF][HW][P16]
    __bf16* Ws = h16_lds + (size_t)NF * HW * H32_P16;           // [3][32][P16]
    int bid = blockIdx.x;
    const int t1i = bid % tiles1; bid /= tiles1;
    const int t0i = bid % tiles0; const int bi = bid / tiles0;
    const int o0b = t0i * H32_TO0, o1b = t1i * H32_TO1;
    const int lo0 = g.d0 < 0 ? -(g.T0 - 1) : 0, lo1 = g.d1 < 0 ? -(g.T1 - 1) : 0;
    const int in0b = o0b + g.off0 + lo0, in1b = o1b * g.s1 + g.off1 + lo1;
    const int total = NF * HW * 8;
    const float* xb = g.x.p + (size_t)bi * g.x.D0 * g.x.P1 * g.x.Cp;
    for (int i = tid; i < total; i += 256) {
        const int c4 = i & 7, pos = i >> 3;
        const int h0 = pos / HW, h1 = pos - h0 * HW;
        const int i0 = in0b + h0, i1 = in1b + h1;
        const bool ok = (unsigned)i0 < (unsigned)g.x.D0 && (unsigned)i1 < (unsigned)g.x.D1;
        const f32x4 v = ok ? ld4(xb + ((size_t)i0 * g.x.P1 + i1) * g.x.Cp + 4 * c4) : zero4();
        *reinterpret_cast<bf16x4*>(Hs + (size_t)pos * H32_P16 + 4 * c4) = to_bf16x4(v);
    }
    const int wn = tid >> 3, wc = 4 * (tid & 7);
    const int NT = g.T0 * g.T1;
    const bool wrow_ok = wn < 16 * TN;          // the weight matrix has 16 TN rows
    const float* wsrc = W + (size_t)(wrow_ok ? wn : 0) * Kp + wc;
    __bf16* wdst = Ws + wn * H32_P16 + wc;
    *reinterpret_cast<bf16x4*>(wdst) = to_bf16x4(wrow_ok ? ld4(wsrc) : zero4());
    if (NT > 1) *reinterpret_cast<bf16x4*>(wdst + 32 * H32_P16) = to_bf16x4(wrow_ok ? ld4(wsrc + 32) : zero4());
    f32x4 wreg = (NT > 2 && wrow_ok) ? ld4(wsrc + 64) : zero4();
    __syncthreads();
    f32x4 acc[TN][2];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = zero4();
    int t0 = 0, t1 = 0, slot = 0;
    for (int tap = 0; tap < NT; ++tap) {
        f32x4 wnew = zero4();
        if (tap + 3 < NT && wrow_ok) wnew = ld4(wsrc + (tap + 3) * 32);
        const __bf16* hrow = Hs + ((size_t)(wave + g.d0 * t0 - lo0) * HW + (g.d1 * t1 - lo1)) * H32_P16 + 8 * lg;
        const __bf16* wrow = Ws + slot * 32 * H32_P16 + l15 * H32_P16 + 8 * lg;
        bf16x8 af[2], wf[TN];
#pragma unroll
        for (int b = 0; b < 2; ++b) af[b] = *reinterpret_cast<const bf16x8*>(hrow + (size_t)((16 * b + l15) * g.s1) * H32_P16);
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = *reinterpret_cast<const bf16x8*>(wrow + 16 * a * H32_P16);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], af[b], acc[a][b], 0, 0, 0);
        const int wslot = slot == 0 ? 2 : slot - 1;
        if (tap + 2 < NT) *reinterpret_cast<bf16x4*>(wdst + wslot * 32 * H32_P16) = to_bf16x4(wreg);
        if (tap + 1 < NT) __syncthreads();
        wreg = wnew;
        slot = slot == 2 ? 0 : slot + 1;
        if (++t1 == g.T1) { t1 = 0; ++t0; }
    }
    const int o0 = o0b + wave;
    if (o0 >= g.O0) return;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int o1 = o1b + 16 * b + l15;
        if (o1 >= g.O1) continue;
        const int m = (bi * g.O0 + o0) * g.O1 + o1;
#pragma unroll
        for (int a = 0; a < TN; ++a) ep.store(m, 16 * a + 4 * lg, acc[a][b], 0);
    }
}

inline Halo32 make_halo32(const ConvSU& l, int M, int Np, int Kp) {
    Halo32 h{l.x, l.g.T0, l.g.T1, l.g.s1, 1, 1, -l.g.p0, -l.g.p1, l.g.O0, l.g.O1, 0, 0, 0};
    h.B = M / std::max(1, l.g.O0 * l.g.O1);
    h.ok = Np == 32 && l.x.Cp == 32 && l.g.s0 == 1 && Kp == l.g.T0 * l.g.T1 * 32 && h.B * l.g.O0 * l.g.O1 == M;
    return h;
}
inline Halo32 make_halo32(const ConvTS& l, int M, int Np, int Kp) {             // dX of a stride-1 layer: rows walk the INPUT map of the layer
    Halo32 h{l.y, l.g.T0, l.g.T1, 1, -1, -1, l.g.p0, l.g.p1, l.D0, l.D1, 0, 0, 0};
    h.B = M / std::max(1, l.D0 * l.D1);
    const bool geom = l.y.Cp == 32 && l.g.s0 == 1 && l.g.s1 == 1 && Kp == l.g.T0 * l.g.T1 * 32 && h.B * l.D0 * l.D1 == M;
    h.ok = Np == 32 && geom; h.ok16 = Np == 16 && geom;
    return h;
}
inline Halo32 make_halo32(const ConvTSP& l, int M, int Np, int Kp) {            // dX of one residue class of a strided layer
    Halo32 h{l.y, l.g.n0, l.g.n1, 1, -1, -1, l.g.c0, l.g.c1, l.g.Q0, l.g.Q1, 0, 0, 0};
    h.B = M / std::max(1, l.g.Q0 * l.g.Q1);
    h.ok = Np == 32 && l.y.Cp == 32 && l.g.n0 >= 1 && l.g.n1 >= 1 && Kp == l.g.n0 * l.g.n1 * 32 && h.B * l.g.Q0 * l.g.Q1 == M;
    return h;
}
inline Halo32 make_halo32(const PlainA&, int, int, int) { Halo32 h{}; h.ok = 0; return h; }

#ifdef ESCX_EXPERIMENTAL
template <class Epi>
inline bool launch_conv32_halo(const Halo32& h, const float* W, int Kp, const Epi& ep, hipStream_t st) {
    if (!h.ok || h.T0 > 4 || h.T1 > 9) return false;
    const int NF = H32_TO0 + h.T0 - 1, HW = (H32_TO1 - 1) * h.s1 + h.T1;
    const size_t lds = ((size_t)NF * HW * H32_P + 3 * 32 * H32_P) * sizeof(float);
    if (lds > 150 * 1024) return false;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv32_halo_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);   // per launch: per-device attribute, no shared flag
    const int tiles0 = (h.O0 + H32_TO0 - 1) / H32_TO0, tiles1 = (h.O1 + H32_TO1 - 1) / H32_TO1;
    hipLaunchKernelGGL((conv32_halo_kernel<Epi>), dim3((unsigned)((size_t)h.B * tiles0 * tiles1)), dim3(256), lds, st, h, W, Kp, tiles0, tiles1, ep);
    return true;
}

#endif
template <class Epi>
inline bool launch_conv32_halo_bf16(const Halo32& h, const float* W, int Kp, const Epi& ep, hipStream_t st) {
    if ((!h.ok && !h.ok16) || h.T0 > 4 || h.T1 > 9) return false;
    const int NF = H32_TO0 + h.T0 - 1, HW = (H32_TO1 - 1) * h.s1 + h.T1;
    const size_t lds = ((size_t)NF * HW * H32_P16 + 3 * 32 * H32_P16) * sizeof(__bf16);       // <= 36 KB
    const int tiles0 = (h.O0 + H32_TO0 - 1) / H32_TO0, tiles1 = (h.O1 + H32_TO1 - 1) / H32_TO1;
    if (h.ok) hipLaunchKernelGGL((conv32_halo_bf16_kernel<2, Epi>), dim3((unsigned)((size_t)h.B * tiles0 * tiles1)), dim3(256), lds, st, h, W, Kp, tiles0, tiles1, ep);
    else hipLaunchKernelGGL((conv32_halo_bf16_kernel<1, Epi>), dim3((unsigned)((size_t)h.B * tiles0 * tiles1)), dim3(256), lds, st, h, W, Kp, tiles0, tiles1, ep);
    return true;
}

}  // namespace escx
