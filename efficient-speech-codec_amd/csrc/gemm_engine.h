// fp32 MFMA GEMM engine for gfx950 (MI355X):  Out = epilogue( A_loader(m, k) . W[n][k] )
//
// Every dense contraction of the ESC hot path (QKV / proj / MLP / merge / split linears, the 5x5 and 3x3
// de-embedding convolutions as implicit GEMMs, the windowed DFT and inverse DFT, the quantiser down/up
// projections) is this one kernel with a pluggable A-side loader and a pluggable epilogue.
//
//  * arithmetic: v_mfma_f32_16x16x4_f32 -- exact fp32 (bitwise an fmaf chain), 157 TFLOP/s peak.  The
//    emitted code indices must be bit-exact against the fp32 reference, so no bf16/fp16/xf32 anywhere.
//  * operands are swapped on purpose: the WEIGHT tile is the MFMA "A" operand and the ACTIVATION tile the
//    "B" operand, so that a lane ends up holding 4 consecutive output features of one output row and
//    the epilogue is a single 16-byte store per lane (bias / residual loads are 16-byte too).
//  * K and N are padded to multiples of 16 in every internal layout (weights are packed once on the host
//    with zero padding), which keeps every global access 16-byte aligned and the MFMA tiles full.
//  * 256 threads = 4 waves stacked along M; each wave owns (BM/4) x BN of the block tile.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

#include "tune_env.h"
#include "launch_prof.h"

// tuning builds (-DESCX_SMALL_PRIO=n): static wave priority of the short, latency-bound launches (GEMM engine, merge / split, PVQ search, combine,
// de-embedding) against the MFMA-dense fused MLP / attention launches of the other batch part they share the GPU with
#ifdef ESCX_SMALL_PRIO
#define ESCX_SET_PRIO_SMALL() __builtin_amdgcn_s_setprio(ESCX_SMALL_PRIO)
#else
#define ESCX_SET_PRIO_SMALL()
#endif

namespace escx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// ------------------------------------------------------------------------------------------------
// An epilogue with `static constexpr bool ROWWISE = true` gets the accumulators of its wave instead of per-fragment store() calls:
//   template <int TN, int TM> void finish(f32x4 (&acc)[TN][TM], int wave_row0, int lane, int M) const
// lane (l15, lg) holds D[row = wave_row0 + 16*b + l15][col = 16*a + 4*lg + r] in acc[a][b][r].  The launch must use one workgroup column.
template <class E, class = void> struct epi_is_rowwise : std::false_type {};
template <class E> struct epi_is_rowwise<E, std::enable_if_t<E::ROWWISE>> : std::true_type {};

template <int BM, int BN, int BK, class Loader, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel(Loader ld, const float* __restrict__ Wt, int M, int Np,
                                                   int Kp, int k_per_z, Epi ep) {
    ESCX_SET_PRIO_SMALL();
    static_assert(BM % 64 == 0 && BN % 16 == 0 && BK % 16 == 0, "tile shape");
    constexpr int LDS_LD = BK + 4;             // +1 access width: rows land on different bank groups
    constexpr int TM = BM / 64;                // 16-row MFMA tiles per wave along M
    constexpr int TN = BN / 16;
    constexpr int KV = BK / 4;                 // float4 per tile row
    constexpr int A4 = BM * KV;
    constexpr int B4 = BN * KV;
    constexpr int AJ = (A4 + 255) / 256;

    __shared__ float As[BM * LDS_LD];
    __shared__ float Bs[BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * k_per_z;
    const int kend = min(Kp, kbeg + k_per_z);

    typename Loader::Ctx ctx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int i = tid + j * 256;
        ctx[j] = ld.make_ctx(m0 + (i < A4 ? i / KV : 0));
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = zero4();

    // Global -> register -> LDS with the NEXT K step's loads in flight while the current one is on the MFMA: the skinny, many-step
    // GEMMs this engine serves (PVQ projections, STFT) are bound by one memory round trip per K step otherwise.
    constexpr int BJ = (B4 + 255) / 256;
    f32x4 ra[AJ], rb[BJ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            ra[j] = (A4 % 256 == 0 || i < A4) ? ld.load4(ctx[j], k0, 4 * (i % KV)) : zero4();
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256;
            const int row = i / KV, c4 = i % KV;
            rb[j] = ((B4 % 256 == 0 || i < B4) && n0 + row < Np) ? ld4(Wt + (size_t)(n0 + row) * Kp + k0 + 4 * c4) : zero4();
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + j * 256;
            if (A4 % 256 == 0 || i < A4) st4(&As[(i / KV) * LDS_LD + 4 * (i % KV)], ra[j]);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + j * 256;
            if (B4 % 256 == 0 || i < B4) st4(&Bs[(i / KV) * LDS_LD + 4 * (i % KV)], rb[j]);
        }
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b) af[b] = ld4(&As[(wave * (BM / 4) + b * 16 + l15) * LDS_LD + kk + 4 * lg]);
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[a] = ld4(&Bs[(a * 16 + l15) * LDS_LD + kk + 4 * lg]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a][r], af[b][r], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // row epilogues (Epi::ROWWISE): the workgroup tile spans the whole output row (Np <= BN), so the epilogue can reduce over it in registers
    if constexpr (epi_is_rowwise<Epi>::value) {
        ep.template finish<TN, TM>(acc, m0 + wave * (BM / 4), lane, M);
        return;
    }
    // lane holds D[n = 4*lg + r][m = l15] of every 16x16 tile: 4 consecutive output features of one row
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wave * (BM / 4) + b * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + a * 16 + 4 * lg;
            if (n < Np) ep.store(m, n, acc[a][b], blockIdx.z);
        }
    }
}

// Exact n / d for 0 <= n < 2^31 with a host-computed multiplier (Granlund-Montgomery): one v_mul_hi + shift instead of the ~25
// VALU instructions of a runtime integer division.  The gather loaders and scatter epilogues divide by runtime extents for
// every 16-byte access; with K = 16..96 that index arithmetic was 10-17 VALU instructions per MFMA of the PVQ projections.
struct FastDiv {
    unsigned mul, shr; int d;
    FastDiv() : mul(0), shr(0), d(1) {}
    explicit FastDiv(int dv) : mul(0), shr(0), d(dv) {
        if (dv > 1) {
            int lg = 0; while ((1ll << lg) < dv) ++lg;                  // ceil(log2 d)
            const int p = 31 + lg;
            mul = (unsigned)(((1ull << p) + (unsigned long long)dv - 1) / (unsigned long long)dv);
            shr = (unsigned)(p - 32);
        }
    }
    __host__ __device__ __forceinline__ int div(int n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return d == 1 ? n : (int)(__umulhi((unsigned)n, mul) >> shr);
#else
        return d == 1 ? n : (int)((((unsigned long long)(unsigned)n * mul) >> 32) >> shr);
#endif
    }
};

// ------------------------------------------------------------------------------------------------
// A-side loaders.  make_ctx(m) is evaluated once per thread-row before the K loop; load4(ctx, k0, kin)
// returns A[m][k0+kin .. +3] (k0 is block-uniform, kin the offset inside the BK tile).
// ------------------------------------------------------------------------------------------------
struct PlainA {                 // A[m][k], row stride lda (multiple of 4 floats)
    const float* A; int lda; int M;
    typedef const float* Ctx;
    __device__ __forceinline__ Ctx make_ctx(int m) const { return m < M ? A + (size_t)m * lda : nullptr; }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const { return c ? ld4(c + k0 + kin) : zero4(); }
};

struct ConvA {                  // implicit-GEMM 'same' convolution over a (D0, D1) grid of Cp-wide tokens
    const float* x; int D0, D1, Cp, T0, T1, M;      // k = (t0*T1 + t1)*Cp + c ; requires BK | Cp
    struct Ctx { int base; int i0; int i1; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.base = -1; c.i0 = 0; c.i1 = 0;
        if (m < M) { const int b = m / (D0 * D1); const int r = m - b * D0 * D1; c.i0 = r / D1; c.i1 = r - c.i0 * D1; c.base = b * D0 * D1; }
        return c;
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        const int tap = k0 / Cp, cc = k0 - tap * Cp + kin;
        const int t0 = tap / T1, t1 = tap - t0 * T1;
        const int j0 = c.i0 + t0 - T0 / 2, j1 = c.i1 + t1 - T1 / 2;
        if (c.base < 0 || j0 < 0 || j0 >= D0 || j1 < 0 || j1 >= D1) return zero4();
        return ld4(x + ((size_t)(c.base + j0 * D1 + j1)) * Cp + cc);
    }
};

struct FrameA {                 // STFT framing with reflect padding: A[(b,f)][k] = wave[b][reflect(f*hop + k + off)]
    const float* wave; int L, T, hop, off, M; FastDiv dT;
    struct Ctx { const float* w; int start; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.w = nullptr; c.start = 0;
        if (m < M) { const int b = dT.div(m), f = m - b * T; c.w = wave + (size_t)b * L; c.start = f * hop + off; }
        return c;
    }
    __device__ __forceinline__ float at(const float* w, int p) const {
        if (p < 0) p = -p;
        if (p >= L) p = 2 * (L - 1) - p;
        return w[p];
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (!c.w) return zero4();
        const int p = c.start + k0 + kin;
        f32x4 v; v[0] = at(c.w, p); v[1] = at(c.w, p + 1); v[2] = at(c.w, p + 2); v[3] = at(c.w, p + 3);
        return v;
    }
};

struct PatchA {                 // PatchEmbed gather: A[(b,ph,pw)][k=(c,df,dt)] = spec[b][pt*pw+dt][c*Fp + pf*ph+df]
    const float* spec; int T, ldf, Fp, H, W, pf, pt, K, M;   // ldf = in_dim*Fp (row stride of a frame)
    struct Ctx { const float* p; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.p = nullptr;
        if (m < M) { const int b = m / (H * W); const int r = m - b * H * W; const int ph = r / W, pw = r - ph * W;
                     c.p = spec + ((size_t)b * T + (size_t)pt * pw) * ldf + pf * ph; }
        return c;
    }
    __device__ __forceinline__ float one(const float* p, int k) const {
        if (k >= K) return 0.f;
        const int c = k / (pf * pt); const int r = k - c * pf * pt; const int df = r / pt, dt = r - df * pt;
        return p[(size_t)dt * ldf + c * Fp + df];
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        if (!c.p) return zero4();
        f32x4 v; const int k = k0 + kin;
        v[0] = one(c.p, k); v[1] = one(c.p, k + 1); v[2] = one(c.p, k + 2); v[3] = one(c.p, k + 3);
        return v;
    }
};

struct ResidualGatherA {        // PVQ framing of (enc - dec): A[(b,t)][k=(o,h,c)] ; requires BK | Cp
    const float* enc; const float* dec; int Hq, W, Cp, Tq, ov, M; FastDiv dTq, dCp, dHq;
    typedef int Ctx;            // element offset of token (b, h=0, w=ov*t), or -1
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        if (m >= M) return -1;
        const int b = dTq.div(m), t = m - b * Tq;
        return (b * Hq * W + ov * t) * Cp;
    }
    __device__ __forceinline__ f32x4 load4(Ctx c, int k0, int kin) const {
        if (c < 0) return zero4();
        const int oh = dCp.div(k0), cc = k0 - oh * Cp + kin;
        const int o = dHq.div(oh), h = oh - o * Hq;
        const size_t idx = (size_t)c + (size_t)(h * W + o) * Cp + cc;
        f32x4 v = ld4(enc + idx);
        if (dec) v -= ld4(dec + idx);
        return v;
    }
};

struct CodeGatherA {            // PVQ de-quantisation: A[(b,t)][g*dt + j] = codebook_g[code[b,g,t]][j]
    const long long* codes; long long bstride; const float* cb; int G, Ksz, dt, Tq, M; FastDiv dTq, ddt;
    struct Ctx { const long long* c; };
    __device__ __forceinline__ Ctx make_ctx(int m) const {
        Ctx c; c.c = nullptr;
        if (m < M) { const int b = dTq.div(m), t = m - b * Tq; c.c = codes + (size_t)b * bstride + t; }
        return c;
    }
    __device__ __forceinline__ f32x4 load4(const Ctx& c, int k0, int kin) const {
        const int k = k0 + kin; const int g = ddt.div(k);
        if (!c.c || g >= G) return zero4();
        long long code = c.c[(size_t)g * Tq];
        code = code < 0 ? 0 : (code >= Ksz ? Ksz - 1 : code);     // a corrupt index must not read outside the codebook (F.embedding would raise)
        return ld4(cb + ((size_t)g * Ksz + (size_t)code) * dt + (k - g * dt));
    }
};

// ------------------------------------------------------------------------------------------------
// Epilogues: store(m, n, v, z) with v = 4 consecutive output features n..n+3 of row m.
// ------------------------------------------------------------------------------------------------
// Cross-lane butterflies over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48) on the VALU: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange half-rows between two registers, so one swap + one op replaces a
// ds_bpermute round trip through the LDS pipe (~100 cycles of latency per step in the softmax / LayerNorm chains).
__device__ __forceinline__ float sum_xor16(float v) {
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float vmax(float a, float b) {       // plain v_max_f32: fmaxf on bit-cast values costs two extra canonicalising ops
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_xor16(float v) {
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float max_xor32(float v) {
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float sum_groups(float v) { return sum_xor32(sum_xor16(v)); }
__device__ __forceinline__ float max_groups(float v) { return max_xor32(max_xor16(v)); }
// value of lane l^32 / l^8
__device__ __forceinline__ float lane_xor32(float v, bool upper_half) {
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(upper_half ? q[0] : q[1]);
}
__device__ __forceinline__ float lane_xor8(float v) {           // DPP row_ror:8 inside each 16-lane row
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xf, 0xf, true));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Branch-free fp32 erf, max error 1.1 ulp over the whole range (fitted and checked in fp64; tests/test_gpu_parity.py
// re-checks it on the device).  libm's erff is a tree of data-dependent branches, which splits the fused MLP's
// inner loop into basic blocks and stops the scheduler from hiding the VALU work under the MFMAs; this form is
// ~25 straight-line VALU ops.   |x| <= 0.92: x + x*P6(x^2) ;  else: sign(x) * (1 - exp(Q8(min(|x|,4)))), Q8 ~ log(erfc).
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float erf_bf(float x) {
    const float t = fabsf(x), s = x * x;
    float a = 8.404849359067157e-05f;
    a = fmaf(a, s, -0.0008151340298354626f);
    a = fmaf(a, s, 0.0052018375135958195f);
    a = fmaf(a, s, -0.026859745383262634f);
    a = fmaf(a, s, 0.11283702403306961f);
    a = fmaf(a, s, -0.37612634897232056f);
    a = fmaf(a, s, 0.12837916612625122f);
    a = fmaf(a, x, x);
    const float u = fminf(t, 4.0f);
    float r = 1.612904043213348e-06f;
    r = fmaf(r, u, -4.556713975034654e-05f);
    r = fmaf(r, u, 0.0005925663863308728f);
    r = fmaf(r, u, -0.004739245865494013f);
    r = fmaf(r, u, 0.026361873373389244f);
    r = fmaf(r, u, -0.10997311770915985f);
    r = fmaf(r, u, -0.6319426894187927f);
    r = fmaf(r, u, -1.130163311958313f);
    r = fmaf(r, u, 0.00030238047474995255f);
    const float b = copysignf(1.0f - exp_fast(r), x);
    return t <= 0.92f ? a : b;
}
// Exact-erf GELU of the fused kernels, 11 straight-line VALU ops (the erf_bf form above costs 29, and at C <= 96 the fused MLP
// issues more VALU than the matrix pipe can hide):  gelu(x) = x*Phi(x) = max(x, 0) - 0.5*|x| * erfc(|x|/sqrt2), with
// erfc(a/sqrt2) = 2^Q8(a) fitted on [0, 5.8] (weighted minimax on the GELU's absolute error, fp64; beyond 5.8 the term is < 4e-8).
// |gelu_bf - gelu| <= 1.25 * 2^-24 * max(|x|, 1) over the whole line - the 0.5*x*(1 + erff(x/sqrt2)) evaluation of the
// reference measures 1.8 on the same scale (1 + erf loses the tail to rounding); escx_test_math / test_device_math re-check both.
__device__ __forceinline__ float gelu_bf(float x) {
    const float a = fabsf(x);       // no clamp: Q8 keeps falling beyond 5.8 (leading coefficient < 0), so the tail term only shrinks
    float r = -1.6904631365832756e-06f;
    r = fmaf(r, a, 2.5084045773837715e-05f);
    r = fmaf(r, a, -0.0001144662601291202f);
    r = fmaf(r, a, -0.0003233331080991775f);
    r = fmaf(r, a, 0.007333371322602034f);
    r = fmaf(r, a, -0.052714187651872635f);
    r = fmaf(r, a, -0.4591154456138611f);
    r = fmaf(r, a, -1.151123285293579f);
    r = fmaf(r, a, 1.126017423302983e-06f - 1.0f);             // the factor 1/2 of 0.5*|x|*erfc rides in the exponent
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(r), fmaxf(x, 0.0f));      // |x| and the negation are operand modifiers: 11 ops
}

struct EpiStore {               // out[m][n] = v (+ bias[n])
    float* out; int ldo; const float* bias;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        if (bias) v += ld4(bias + n);
        st4(out + (size_t)m * ldo + n, v);
    }
};

struct EpiGelu {                // out = gelu(v + bias)   (exact erf form, nn.GELU default)
    float* out; int ldo; const float* bias;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        v += ld4(bias + n);
        v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
        st4(out + (size_t)m * ldo + n, v);
    }
};

struct EpiResidual {            // out = res + (v + bias)   (in-place safe: out may alias res)
    float* out; int ldo; const float* bias; const float* res;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        v += ld4(bias + n);
        st4(out + (size_t)m * ldo + n, ld4(res + (size_t)m * ldo + n) + v);
    }
};

struct EpiQkv {                 // out = (v + bias), q columns (n < nq) additionally * scale  (attention.py:225)
    float* out; int ldo; const float* bias; int nq; float scale;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        v += ld4(bias + n);
        if (n < nq) v *= scale;
        st4(out + (size_t)m * ldo + n, v);
    }
};

struct EpiProjScatter {         // window reverse + un-roll + crop + residual: out[tok] = shortcut[tok] + (v + bias)
    float* out; const float* shortcut; const float* bias; const int* map; int slots, tokens, Cp;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const int b = m / slots, s = m - b * slots;
        const int tok = map[s];
        if (tok < 0) return;
        const size_t idx = ((size_t)b * tokens + tok) * Cp + n;
        v += ld4(bias + n);
        st4(out + idx, ld4(shortcut + idx) + v);
    }
};

struct EpiSplit {               // PatchSplit pixel shuffle (2,1): out[(b, 2h+s, w)][c] = v[(b,h,w)][s*C2p + c]
    float* out; int H, W, C2p;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const int b = m / (H * W); const int r = m - b * H * W; const int h = r / W, w = r - h * W;
        const int s = n / C2p, c = n - s * C2p;
        st4(out + ((size_t)(b * 2 * H + 2 * h + s) * W + w) * C2p + c, v);
    }
};

struct EpiDeembed1 {            // conv5x5 bias + pixel shuffle (pf,pt) into a TIME-major (b, t, f, Cp) map
    float* out; const float* bias; int H, W, Cp, pf, pt;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const int b = m / (H * W); const int r = m - b * H * W; const int h = r / W, w = r - h * W;
        const int q = n / Cp, c = n - q * Cp;
        const int s1 = q / pt, s2 = q - s1 * pt;
        const int f = pf * h + s1, t = pt * w + s2;
        v += ld4(bias + n);
        st4(out + (((size_t)b * (pt * W) + t) * (pf * H) + f) * Cp + c, v);
    }
};

struct EpiSpec {                // conv3x3 -> frame-major spectrum: rows m = (b, t, f); out[(b,t)][ch*Fp + f]
    float* out; const float* bias; int T, F, Fp, in_dim;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        if (n >= in_dim) return;
        const int bt = m / F, f = m - bt * F;
        float* o = out + (size_t)bt * (in_dim * Fp) + f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < in_dim) o[(n + r) * Fp] = v[r] + bias[n + r];
    }
};

struct EpiDeembedC {            // composed de-embedding: n = (cout, s1, s2) -> frame-major spectrum [(b, pt*w+s2)][cout*Fp + pf*h+s1]
    float* out; const float* bias; int H, W, pf, pt, in_dim, Fp;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const int b = m / (H * W); const int r0 = m - b * H * W; const int h = r0 / W, w = r0 - h * W;
        const int Q = pf * pt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = n + r;
            if (nn >= in_dim * Q) continue;
            const int co = nn / Q, q = nn - co * Q, s1 = q / pt, s2 = q - s1 * pt;
            out[((size_t)(b * (pt * W) + pt * w + s2) * in_dim + co) * Fp + pf * h + s1] = v[r] + bias[nn];
        }
    }
};

struct EpiPvqAdd {              // un-frame + post_fuse: out[(b,h,ov*t+o)][c] = dec[...] + v   (csrvq.py:19-21)
    float* out; const float* dec; int Hq, W, Cp, Tq, ov; FastDiv dTq, dCp, dHq;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int) const {
        const int b = dTq.div(m), t = m - b * Tq;
        const int oh = dCp.div(n), c = n - oh * Cp;
        const int o = dHq.div(oh), h = oh - o * Hq;
        const size_t idx = ((size_t)(b * Hq + h) * W + ov * t + o) * Cp + c;
        if (dec) v += ld4(dec + idx);
        st4(out + idx, v);
    }
};

struct EpiPartial {             // split-K partial sums, reduced in a fixed order by the consumer (deterministic)
    float* out; int M, Np;
    __device__ __forceinline__ void store(int m, int n, f32x4 v, int z) const {
        st4(out + ((size_t)z * M + m) * Np + n, v);
    }
};

// ------------------------------------------------------------------------------------------------
// Host-side tile selection and launch.
// ------------------------------------------------------------------------------------------------
inline int pick_bk(int Kp) {
    static const int env_bk = [] { const char* e = ESCX_TUNE_ENV("ESCX_BK"); return e ? atoi(e) : 0; }();          // tuning aid (results are identical for every step size)
    if (env_bk > 0 && Kp % env_bk == 0) return env_bk;
    return Kp % 48 == 0 ? 48 : (Kp % 80 == 0 ? 80 : (Kp % 32 == 0 ? 32 : 16));
}     // 80: the C = 72 maps (a 16-wide step there means 5x the K steps)
inline int pick_bn(int Np) {
    // padded width, with a mild preference for wide tiles (every N tile re-stages the A tile)
    const int cands[4] = {96, 48, 32, 16};
    int best = 16; double best_cost = 1e30;
    for (int c : cands) { const double cost = (double)((Np + c - 1) / c * c) * (1.0 + 8.0 / c); if (cost < best_cost) { best_cost = cost; best = c; } }
    return best;
}

template <int BM, int BN, int BK, class Loader, class Epi>
inline void launch_tile(const Loader& ld, const float* Wt, int M, int Np, int Kp, int splits, const Epi& ep, hipStream_t s) {
    int kIters = Kp / BK;
    int per = (kIters + splits - 1) / splits;
    dim3 grid((M + BM - 1) / BM, (Np + BN - 1) / BN, (kIters + per - 1) / per);
    ESCX_LAUNCH((gemm_kernel<BM, BN, BK, Loader, Epi>), grid, dim3(256), 0, s, ld, Wt, M, Np, Kp, per * BK, ep);
}

template <int BM, int BK, class Loader, class Epi>
inline void launch_bn(int BN, const Loader& ld, const float* Wt, int M, int Np, int Kp, int splits, const Epi& ep, hipStream_t s) {
    switch (BN) {
        case 96: launch_tile<BM, 96, BK>(ld, Wt, M, Np, Kp, splits, ep, s); break;
        case 48: launch_tile<BM, 48, BK>(ld, Wt, M, Np, Kp, splits, ep, s); break;
        case 32: launch_tile<BM, 32, BK>(ld, Wt, M, Np, Kp, splits, ep, s); break;
        default: launch_tile<BM, 16, BK>(ld, Wt, M, Np, Kp, splits, ep, s); break;
    }
}

// Generic entry: picks BN from Np and BK from Kp (or uses the caller's BK when a loader constrains it).
template <int BM, class Loader, class Epi>
inline void launch_gemm(const Loader& ld, const float* Wt, int M, int Np, int Kp, const Epi& ep, hipStream_t s,
                        int splits = 1, int force_bk = 0) {
    const int BN = pick_bn(Np);
    const int BK = force_bk ? force_bk : pick_bk(Kp);
    switch (BK) {
        case 80: launch_bn<BM, 80>(BN, ld, Wt, M, Np, Kp, splits, ep, s); break;
        case 48: launch_bn<BM, 48>(BN, ld, Wt, M, Np, Kp, splits, ep, s); break;
        case 32: launch_bn<BM, 32>(BN, ld, Wt, M, Np, Kp, splits, ep, s); break;
        default: launch_bn<BM, 16>(BN, ld, Wt, M, Np, Kp, splits, ep, s); break;
    }
}

}  // namespace escx
