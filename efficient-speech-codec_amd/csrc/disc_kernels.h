// Device code of the adversarial step (BASELINE configs[4]): the DAC discriminator of the reference
// (esc/models/discriminator.py:31-221: multi-period + multi-resolution-spectrogram discriminators, weight-normalised convolutions with
// LeakyReLU(0.1)) and the least-squares GAN / feature-matching losses (esc/modules/loss/gan_loss.py:5-51).
//
// Every convolution is an implicit GEMM on the fp32 MFMA engine of gemm_engine.h over channels-last activations:
//   tensor view  T(b, i0, i1, c) = p[((b * D0 + i0) * P1 + i1) * Cp + c]      (P1 >= D1: row pitch, so that the five band stacks of an MRD
//                                                                            write straight into their slice of the concatenated map)
//   forward      ConvS loader (strided taps, zero padding) x packed weight-normalised weights, epilogue bias + LeakyReLU
//   dX           ConvTS loader (the transposed, strided gather over the output gradient) x transposed weights, accumulating epilogue
//   dW, db       gemm_dw_kernel (train_kernels.h) with the output-gradient rows and the ConvS gather as operands
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"

namespace escx {

struct TView { float* p; int D0, D1, P1, Cp; };           // see above; batch stride = D0 * P1 * Cp

struct ConvGeom { int T0, T1, s0, s1, p0, p1, O0, O1; };  // taps, strides, pads, output extent

// Loader cost matters here: the engine calls load4 twice per thread per 16-wide K step, next to 48 MFMAs per wave.  The straightforward form (tap and channel
// from k by two divisions, a 64-bit (b, i0, i1, c) address product) is ~30 VALU instructions per call = 15 % of the MFMA time of the 1024-channel layers.
// Here the row context carries a 32-bit element offset of the row's tap-(0, 0) position (wrapping arithmetic: out-of-range taps are masked by the bounds
// test), and the tap index is derived from k0 ALONE - in the engine k0 is wave-uniform, so the two divisions and the tap offset run on the scalar unit;
// a K step that straddles taps (Cp < 16: the one-channel input layers) takes the general path.  Feature maps hold < 2^32 elements (checked by the host).

// forward gather: A[(b, o0, o1)][k = (t0*T1 + t1)*Cp + c] = X(b, o0*s0 + t0 - p0, o1*s1 + t1 - p1, c)
struct ConvS {
    TView x; ConvGeom g; int M; FastDiv dO, dO1, dCp, dT1;
    int fast;                   // set by conv_gemm: the engine's K step divides the channel count, so no step straddles a tap (ConvSU only)
    struct Ctx { int ok, i0, i1; unsigned base; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.ok = 0; c.i0 = 0; c.i1 = 0; c.base = 0;
        if (m < M) {
            const int b = dO.div(m); const int r = m - b * g.O0 * g.O1; const int o0 = dO1.div(r), o1 = r - o0 * g.O1;
            c.ok = 1; c.i0 = o0 * g.s0 - g.p0; c.i1 = o1 * g.s1 - g.p1;
            c.base = (((unsigned)b * (unsigned)x.D0 + (unsigned)c.i0) * (unsigned)x.P1 + (unsigned)c.i1) * (unsigned)x.Cp;
        }
        return c;
    }
    __device__ __forceinline__ f32x4 at(const Ctx& c, int tap, int cc) const {
        if (tap >= g.T0 * g.T1) return zero4();
        const int t0 = dT1.div(tap), t1 = tap - t0 * g.T1;
        if ((unsigned)(c.i0 + t0) >= (unsigned)x.D0 || (unsigned)(c.i1 + t1) >= (unsigned)x.D1) return zero4();
        return ld4(x.p + (c.base + (unsigned)((t0 * x.P1 + t1) * x.Cp) + (unsigned)cc));
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (!c.ok) return zero4();
        const int tap = dCp.div(k0);
        const int cc = k0 - tap * x.Cp + kin;
        if (cc < x.Cp) return at(c, tap, cc);
        const int e = dCp.div(cc);
        return at(c, tap + e, cc - e * x.Cp);
    }
};
// The same gather for callers whose k0 is wave-uniform (the GEMM engine; the dW kernel walks per-thread columns and keeps ConvS): when the channel count is a
// multiple of the 16-wide K step no step straddles a tap, and the load is branch-free - tap, tap offset and tap validity on the scalar unit, per lane two adds,
// two compares and the selects.  Measured on the 1024 -> 1024 period layer: forward 105 -> 110 TFLOP/s (the same form in ConvTS / ConvTSP: dX 98 -> 103, 86 -> 94).
struct ConvSU : ConvS {
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (!fast) return ConvS::load4(c, k0, kin);
        const int tap = dCp.div(k0);
        const int t0 = dT1.div(tap), t1 = tap - t0 * g.T1;
        const unsigned toff = (unsigned)((t0 * x.P1 + t1) * x.Cp + (k0 - tap * x.Cp));
        const bool ok = c.ok && tap < g.T0 * g.T1 && (unsigned)(c.i0 + t0) < (unsigned)x.D0 && (unsigned)(c.i1 + t1) < (unsigned)x.D1;
        const f32x4 v = ld4(x.p + (ok ? c.base + toff + (unsigned)kin : 0u));
        return ok ? v : zero4();
    }
};

// transposed gather for dX: A[(b, i0, i1)][k = (t0*T1 + t1)*Cp + co] = dY(b, (i0 + p0 - t0) / s0, (i1 + p1 - t1) / s1, co) where divisible
struct ConvTS {
    TView y; ConvGeom g; int D0, D1, M; FastDiv dI, dI1, dCp, dT1, dS0, dS1;     // D0, D1: extent of the INPUT map the rows walk
    int fast;                   // set by conv_gemm (see ConvS)
    struct Ctx { int ok, i0, i1; unsigned base; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.ok = 0; c.i0 = 0; c.i1 = 0; c.base = 0;
        if (m < M) {
            const int b = dI.div(m); const int r = m - b * D0 * D1; const int i0 = dI1.div(r);
            c.ok = 1; c.i0 = i0 + g.p0; c.i1 = r - i0 * D1 + g.p1;
            c.base = (unsigned)b * (unsigned)y.D0 * (unsigned)y.P1 * (unsigned)y.Cp;           // of the clip; unit strides add (i0, i1) below
        }
        return c;
    }
    __device__ __forceinline__ f32x4 at(const Ctx& c, int tap, int cc) const {
        if (tap >= g.T0 * g.T1) return zero4();
        const int t0 = dT1.div(tap), t1 = tap - t0 * g.T1;
        const int a0 = c.i0 - t0, a1 = c.i1 - t1;
        if (a0 < 0 || a1 < 0) return zero4();
        int o0 = a0, o1 = a1;
        if (g.s0 != 1 || g.s1 != 1) {                   // (strided layers normally take the residue-class form below)
            o0 = dS0.div(a0); o1 = dS1.div(a1);
            if (o0 * g.s0 != a0 || o1 * g.s1 != a1) return zero4();
        }
        if (o0 >= y.D0 || o1 >= y.D1) return zero4();
        return ld4(y.p + (c.base + (unsigned)((o0 * y.P1 + o1) * y.Cp) + (unsigned)cc));
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (fast && g.s0 == 1 && g.s1 == 1) {                       // the stride-1 layers (the strided ones take the residue-class form): branch-free
            const int tap = dCp.div(k0);
            const int t0 = dT1.div(tap), t1 = tap - t0 * g.T1;
            const int o0 = c.i0 - t0, o1 = c.i1 - t1;
            const bool ok = c.ok && tap < g.T0 * g.T1 && (unsigned)o0 < (unsigned)y.D0 && (unsigned)o1 < (unsigned)y.D1;
            const f32x4 v = ld4(y.p + (ok ? c.base + (unsigned)((o0 * y.P1 + o1) * y.Cp) + (unsigned)(k0 - tap * y.Cp + kin) : 0u));
            return ok ? v : zero4();
        }
        if (!c.ok) return zero4();
        const int tap = dCp.div(k0);
        const int cc = k0 - tap * y.Cp + kin;
        if (cc < y.Cp) return at(c, tap, cc);
        const int e = dCp.div(cc);
        return at(c, tap + e, cc - e * y.Cp);
    }
};

// Strided convolutions: the input positions of one residue class (i0 = r0 + s0*q0, i1 = r1 + s1*q1) receive gradient from a fixed SUBSET of
// the taps (t = tmin + a*s with tmin = (r + p) mod s), so dX is run once per class over compact tap sets instead of once over all taps with
// (s0*s1 - 1)/(s0*s1) of the products structurally zero (the stride-3 layers of the period discriminators are where the FLOPs are).
struct PhaseGeom { int r0, r1, n0, n1, c0, c1, Q0, Q1, s0, s1; };       // n: taps of this class per axis; o = q + c - a; Q: positions of the class
__host__ __device__ inline int phase_tmin(int r, int p, int s) { return (r + p) % s; }
__host__ __device__ inline int phase_ntaps(int r, int p, int s, int T) { const int tm = phase_tmin(r, p, s); return tm < T ? (T - tm + s - 1) / s : 0; }

struct ConvTSP {                // A[(b, q0, q1)][k = (a*n1 + b1)*Cp + co] = dY(b, q0 + c0 - a, q1 + c1 - b1, co)
    TView y; PhaseGeom g; int M; FastDiv dQ, dQ1, dCp, dN1;
    int fast;                   // set by conv_gemm (see ConvS)
    struct Ctx { int ok, q0, q1; unsigned base; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.ok = 0; c.q0 = 0; c.q1 = 0; c.base = 0;
        if (m < M) {
            const int b = dQ.div(m); const int r = m - b * g.Q0 * g.Q1; const int q0 = dQ1.div(r);
            c.ok = 1; c.q0 = q0 + g.c0; c.q1 = r - q0 * g.Q1 + g.c1;
            c.base = (((unsigned)b * (unsigned)y.D0 + (unsigned)c.q0) * (unsigned)y.P1 + (unsigned)c.q1) * (unsigned)y.Cp;
        }
        return c;
    }
    __device__ __forceinline__ f32x4 at(const Ctx& c, int tap, int cc) const {
        if (tap >= g.n0 * g.n1) return zero4();
        const int a = dN1.div(tap), b1 = tap - a * g.n1;
        if ((unsigned)(c.q0 - a) >= (unsigned)y.D0 || (unsigned)(c.q1 - b1) >= (unsigned)y.D1) return zero4();
        return ld4(y.p + (c.base - (unsigned)((a * y.P1 + b1) * y.Cp) + (unsigned)cc));
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (fast) {
            const int tap = dCp.div(k0);
            const int a = dN1.div(tap), b1 = tap - a * g.n1;
            const unsigned toff = (unsigned)((a * y.P1 + b1) * y.Cp - (k0 - tap * y.Cp));
            const bool ok = c.ok && tap < g.n0 * g.n1 && (unsigned)(c.q0 - a) < (unsigned)y.D0 && (unsigned)(c.q1 - b1) < (unsigned)y.D1;
            const f32x4 v = ld4(y.p + (ok ? c.base - toff + (unsigned)kin : 0u));
            return ok ? v : zero4();
        }
        if (!c.ok) return zero4();
        const int tap = dCp.div(k0);
        const int cc = k0 - tap * y.Cp + kin;
        if (cc < y.Cp) return at(c, tap, cc);
        const int e = dCp.div(cc);
        return at(c, tap + e, cc - e * y.Cp);
    }
};

// dX epilogues.  mode 0: out += v.  mode 1 (every position of the map is covered by exactly one dX launch): the FINAL pre-activation
// gradient of the input map in one write, out = (init + v) * LeakyReLU'(y): init = the loss gradient of that map (NULL = 0), y = the map itself
// (NULL = its producer has no activation); init and y share out's layout.  Same arithmetic as view_copy + accumulate + leaky_bwd_kernel.
struct GradOut {
    TView o; const float* init; const float* y; int final_;
    __device__ __forceinline__ void put(size_t off, f32x4 v) const {
        float* q = o.p + off;
        if (!final_) { st4(q, ld4(q) + v); return; }
        if (init) v = ld4(init + off) + v;
        if (y) {
            const f32x4 a = ld4(y + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? v[e] : 0.1f * v[e];
        }
        st4(q, v);
    }
};
struct EpiAccumPhase {          // out(b, r0 + s0*q0, r1 + s1*q1, n..)
    GradOut w; PhaseGeom g; FastDiv dQ, dQ1;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const TView& o = w.o;
        if (n >= o.Cp) return;
        const int b = dQ.div(m), r = m - b * g.Q0 * g.Q1; const int q0 = dQ1.div(r), q1 = r - q0 * g.Q1;
        w.put((((size_t)b * o.D0 + g.r0 + g.s0 * q0) * o.P1 + g.r1 + g.s1 * q1) * o.Cp + n, v);
    }
};

struct ViewRowsA {              // rows m = (b, i0, i1) of a view as a plain matrix (dY operand of the dW contraction)
    TView v; int M; FastDiv dR, dD1;
    typedef const float* Ctx;
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        if (m >= M) return nullptr;
        const int b = dR.div(m), r = m - b * v.D0 * v.D1; const int i0 = dD1.div(r), i1 = r - i0 * v.D1;
        return v.p + (((size_t)b * v.D0 + i0) * v.P1 + i1) * v.Cp;
    }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const { return (c && k0 + kin < v.Cp) ? ld4(c + k0 + kin) : zero4(); }
};

struct EpiConvOut {             // out(b, o0, o1, n..n+3) = act(v + bias[n]) ; act = LeakyReLU(0.1) (discriminator.py:28, 28)
    TView o; const float* bias; int act; FastDiv dR, dD1;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        if (n >= o.Cp) return;
        const int b = dR.div(m), r = m - b * o.D0 * o.D1; const int i0 = dD1.div(r), i1 = r - i0 * o.D1;
        v += ld4(bias + n);
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
        }
        st4(o.p + (((size_t)b * o.D0 + i0) * o.P1 + i1) * o.Cp + n, v);
    }
};

struct EpiAccumView {           // out(b, i0, i1, n..): dX of a layer into the gradient of its input map
    GradOut w; FastDiv dR, dD1;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const TView& o = w.o;
        if (n >= o.Cp) return;
        const int b = dR.div(m), r = m - b * o.D0 * o.D1; const int i0 = dD1.div(r), i1 = r - i0 * o.D1;
        w.put((((size_t)b * o.D0 + i0) * o.P1 + i1) * o.Cp + n, v);
    }
};

// ------------------------------------------------------------------------------------------------
// weight norm (torch.nn.utils.weight_norm, dim 0): w = g * v / ||v||; one workgroup per output channel.
//   v [Cout][Cin][T] (reference layout), Wf [CoutP][Kf] with k = t*CinP + ci, Wt [CinR][Kt] with k = t*CoutP + co
// ------------------------------------------------------------------------------------------------
//   Wp (strided layers only): per residue class (r0, r1) a compact [CinR][n0*n1*CoutP] matrix, classes stored back to back
__global__ __launch_bounds__(256) void wn_pack_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ b,
                                                      float* __restrict__ Wf, float* __restrict__ Wt, float* __restrict__ bias, float* __restrict__ scale,
                                                      int Cout, int Cin, int T, int CinP, int CoutP, int Kf, int Kt, float* __restrict__ Wp, int T0, int T1,
                                                      int s0, int s1, int p0, int p1, int CinR) {
    const int co = blockIdx.x;
    __shared__ float red[256];
    const float* vr = v + (size_t)co * Cin * T;
    float s = 0.f;
    for (int i = threadIdx.x; i < Cin * T; i += 256) s += vr[i] * vr[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    const float nrm = sqrtf(red[0]);
    const float sc = g[co] / nrm;
    if (threadIdx.x == 0) { scale[2 * co] = sc; scale[2 * co + 1] = nrm; bias[co] = b[co]; }
    for (int i = threadIdx.x; i < Cin * T; i += 256) {
        const int ci = i / T, t = i - ci * T;
        const float w = vr[i] * sc;
        Wf[(size_t)co * Kf + t * CinP + ci] = w;
        Wt[(size_t)ci * Kt + t * CoutP + co] = w;
        if (Wp) {
            const int t0 = t / T1, t1 = t - t0 * T1;
            const int r0 = ((t0 - p0) % s0 + s0) % s0, r1 = ((t1 - p1) % s1 + s1) % s1;
            const int a = (t0 - phase_tmin(r0, p0, s0)) / s0, b1 = (t1 - phase_tmin(r1, p1, s1)) / s1;
            size_t off = 0;
            for (int q0 = 0; q0 < s0; ++q0)
                for (int q1 = 0; q1 < s1; ++q1) {
                    if (q0 == r0 && q1 == r1) { q0 = s0; break; }
                    off += (size_t)CinR * phase_ntaps(q0, p0, s0, T0) * phase_ntaps(q1, p1, s1, T1) * CoutP;
                }
            const int n1 = phase_ntaps(r1, p1, s1, T1), kp = phase_ntaps(r0, p0, s0, T0) * n1 * CoutP;
            Wp[off + (size_t)ci * kp + (a * n1 + b1) * CoutP + co] = w;
        }
    }
}
// gradient of (g, v) from the packed weight gradient dWf [CoutP][Kf]:  dg = <dW, v> / ||v|| ;  dv = (g/||v||) (dW - <dW, v> v / ||v||^2)
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dWf, const float* __restrict__ v, const float* __restrict__ scale,
                                                     float* __restrict__ dv, float* __restrict__ dg, int Cin, int T, int CinP, int Kf) {
    const int co = blockIdx.x;
    __shared__ float red[256];
    const float* vr = v + (size_t)co * Cin * T;
    float s = 0.f;
    for (int i = threadIdx.x; i < Cin * T; i += 256) { const int ci = i / T, t = i - ci * T; s += dWf[(size_t)co * Kf + t * CinP + ci] * vr[i]; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    const float dot = red[0], sc = scale[2 * co], nrm = scale[2 * co + 1];
    if (threadIdx.x == 0) dg[co] = dot / nrm;
    const float q = dot / (nrm * nrm);
    for (int i = threadIdx.x; i < Cin * T; i += 256) { const int ci = i / T, t = i - ci * T; dv[(size_t)co * Cin * T + i] = sc * (dWf[(size_t)co * Kf + t * CinP + ci] - q * vr[i]); }
}

// ------------------------------------------------------------------------------------------------
// Discriminator.preprocess (discriminator.py:211-216): y = 0.8 (x - mean) / (max|x - mean| + 1e-9), one workgroup per clip.
// stats[b] = {mean, max|z|, argmax, sign(z_argmax)} for the backward.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void disc_preprocess_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats, int L) {
    const int b = blockIdx.x;
    __shared__ float red[1024]; __shared__ int redi[1024];
    const float* xr = x + (size_t)b * L;
    float s = 0.f;
    for (int i = threadIdx.x; i < L; i += 1024) s += xr[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    const float mean = red[0] / (float)L;
    __syncthreads();
    float mx = -1.f; int mi = 0;
    for (int i = threadIdx.x; i < L; i += 1024) { const float a = fabsf(xr[i] - mean); if (a > mx) { mx = a; mi = i; } }
    red[threadIdx.x] = mx; redi[threadIdx.x] = mi;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if (threadIdx.x < k) {
            const float o = red[threadIdx.x + k]; const int oi = redi[threadIdx.x + k];
            if (o > red[threadIdx.x] || (o == red[threadIdx.x] && oi < redi[threadIdx.x])) { red[threadIdx.x] = o; redi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const float m = red[0]; const int am = redi[0];
    const float sc = 0.8f / (m + 1e-9f);
    for (int i = threadIdx.x; i < L; i += 1024) y[(size_t)b * L + i] = (xr[i] - mean) * sc;
    if (threadIdx.x == 0) { stats[4 * b] = mean; stats[4 * b + 1] = m; stats[4 * b + 2] = (float)am; stats[4 * b + 3] = (xr[am] - mean) >= 0.f ? 1.f : -1.f; }
}
// dx = dz - mean(dz), dz = sc * dy - [sc * <dy, y> / (m + eps)] * sign(z_a) e_a   (y = sc * z, so <dy, z> = <dy, y> / sc)
__global__ __launch_bounds__(1024) void disc_preprocess_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ stats,
                                                                   float* __restrict__ dx, int L) {
    const int b = blockIdx.x;
    __shared__ float r1[1024], r2[1024];
    const float* g = dy + (size_t)b * L; const float* yr = y + (size_t)b * L;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < L; i += 1024) { s1 += g[i]; s2 += g[i] * yr[i]; }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) { if (threadIdx.x < k) { r1[threadIdx.x] += r1[threadIdx.x + k]; r2[threadIdx.x] += r2[threadIdx.x + k]; } __syncthreads(); }
    const float m = stats[4 * b + 1], sgn = stats[4 * b + 3]; const int am = (int)stats[4 * b + 2];
    const float sc = 0.8f / (m + 1e-9f);
    const float spike = r2[0] / (m + 1e-9f) * sgn;          // d/dm of sc * z summed against dy, routed to the argmax element
    const float mean_dz = (sc * r1[0] - spike) / (float)L;
    for (int i = threadIdx.x; i < L; i += 1024) dx[(size_t)b * L + i] = sc * g[i] - (i == am ? spike : 0.f) - mean_dz;
}

// ------------------------------------------------------------------------------------------------
// input builders
// ------------------------------------------------------------------------------------------------
// MPD (discriminator.py:48-57): right reflect-pad to a multiple of the period (a whole period when it already is one), view (L/p, p); Cp = 4
__global__ void mpd_input_kernel(const float* __restrict__ y, float* __restrict__ out, int B, int L, int D0, int p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * D0 * p) return;
    const int pos = (int)(idx % ((long long)D0 * p)); const int b = (int)(idx / ((long long)D0 * p));
    const int src = pos < L ? pos : 2 * (L - 1) - pos;
    st4(out + idx * 4, f32x4{y[(size_t)b * L + src], 0.f, 0.f, 0.f});
}
__global__ void mpd_input_bwd_kernel(const float* __restrict__ din, float* __restrict__ dy, int B, int L, int D0, int p) {      // dy += gather (with the mirrored tail)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L) return;
    const int j = (int)(idx % L), b = (int)(idx / L);
    const long long base = (long long)b * D0 * p;
    float s = din[(base + j) * 4];
    const int q = 2 * (L - 1) - j;                      // padded position that mirrors onto j
    if (q >= L && q < D0 * p) s += din[(base + q) * 4];
    dy[idx] += s;
}
// MRD: band [lo, hi) of the spectrogram rows [B*T][2*Fq] (re | im) -> [B][T][hi-lo][4] = (re, im, 0, 0)   ("b 1 f t c -> (b 1) c t f", :156-158)
__global__ void mrd_band_kernel(const float* __restrict__ spec, float* __restrict__ out, long long rows, int Fq, int lo, int nb) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nb) return;
    const long long r = idx / nb; const int f = lo + (int)(idx - r * nb);
    st4(out + idx * 4, f32x4{spec[r * 2 * Fq + f], spec[r * 2 * Fq + Fq + f], 0.f, 0.f});
}
__global__ void mrd_band_bwd_kernel(const float* __restrict__ din, float* __restrict__ dspec, long long rows, int Fq, int lo, int nb) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nb) return;
    const long long r = idx / nb; const int f = lo + (int)(idx - r * nb);
    const f32x4 g = ld4(din + idx * 4);
    dspec[r * 2 * Fq + f] = g[0]; dspec[r * 2 * Fq + Fq + f] = g[1];
}

// dY <- dY * LeakyReLU'(Y) on a view (Y is the POST-activation map: its sign is the pre-activation's)
__global__ void leaky_bwd_kernel(TView dy, TView y, long long n4, int B) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const int V = y.Cp / 4;
    const int c4 = (int)(idx % V); long long r = idx / V;
    const int i1 = (int)(r % y.D1); r /= y.D1; const int i0 = (int)(r % y.D0), b = (int)(r / y.D0);
    const size_t oy = (((size_t)b * y.D0 + i0) * y.P1 + i1) * y.Cp + 4 * c4, od = (((size_t)b * dy.D0 + i0) * dy.P1 + i1) * dy.Cp + 4 * c4;
    const f32x4 yv = ld4(y.p + oy); f32x4 g = ld4(dy.p + od);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.1f * g[e];
    st4(dy.p + od, g);
}
__global__ void add_into_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
// dst view <- src view (or zero when src.p == nullptr)
__global__ void view_copy_kernel(TView dst, TView src, long long n4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const int V = dst.Cp / 4;
    const int c4 = (int)(idx % V); long long r = idx / V;
    const int i1 = (int)(r % dst.D1); r /= dst.D1; const int i0 = (int)(r % dst.D0), b = (int)(r / dst.D0);
    f32x4 v = zero4();
    if (src.p) v = ld4(src.p + (((size_t)b * src.D0 + i0) * src.P1 + i1) * src.Cp + 4 * c4);
    st4(dst.p + (((size_t)b * dst.D0 + i0) * dst.P1 + i1) * dst.Cp + 4 * c4, v);
}

// ------------------------------------------------------------------------------------------------
// GAN losses on feature-map views (gan_loss.py:30-51).  per clip: mean over the C real channels x D0 x D1.
//   mode 0: (target - x)^2          d/dx = -2 (target - x) / n         (least-squares GAN terms)
//   mode 1: |x - ref|               d/dx = sign(x - ref) / n           (feature matching; ref is detached)
// part[b][block]; gradient written to a view shaped like x (pad channels zero).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gan_term_kernel(TView x, TView ref, TView grad, float* __restrict__ part, int mode, float target, int C,
                                                       int blocks_per_clip, float inv_n, const float* __restrict__ gscale) {
    const int b = blockIdx.y;
    __shared__ float red[256];
    const int V = x.Cp / 4;                                     // four channels per thread and trip (Cp is a multiple of 4)
    const long long per4 = (long long)x.D0 * x.D1 * V;
    const float gs = gscale ? gscale[b] * inv_n : inv_n;         // d (g_b * term_b) / d x: the upstream per-clip gradient folded in (backward-only launches)
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per4; i += (long long)blocks_per_clip * 256) {
        const int c = 4 * (int)(i % V); const long long r = i / V; const int i1 = (int)(r % x.D1); const int i0 = (int)(r / x.D1);
        const size_t ox = (((size_t)b * x.D0 + i0) * x.P1 + i1) * x.Cp + c;
        const f32x4 xv = ld4(x.p + ox);
        f32x4 rv = zero4(), gv = zero4();
        if (mode != 0) rv = ld4(ref.p + (((size_t)b * ref.D0 + i0) * ref.P1 + i1) * ref.Cp + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {                            // same per-element arithmetic as the scalar form (a thread now owns float4 groups)
            if (c + e < C) {
                if (mode == 0) { const float d = target - xv[e]; acc += d * d; gv[e] = -2.f * d * gs; }
                else { const float d = xv[e] - rv[e]; acc += fabsf(d); gv[e] = (d > 0.f ? gs : (d < 0.f ? -gs : 0.f)); }
            }
        }
        if (grad.p) st4(grad.p + (((size_t)b * grad.D0 + i0) * grad.P1 + i1) * grad.Cp + c, gv);
    }
    if (!part) return;
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[(size_t)b * blocks_per_clip + blockIdx.x] = red[0] * inv_n;
}

struct EpiStoreN {              // out[m][n..n+3] = v for n < nvalid (row stride ldo may be smaller than the GEMM's padded width)
    float* out; int ldo, nvalid;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        if (n < nvalid) st4(out + (size_t)m * ldo + n, v);
    }
};

}  // namespace escx
