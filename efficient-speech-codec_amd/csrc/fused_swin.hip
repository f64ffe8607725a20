// Launchers for the fused (wave-autonomous, register-resident) Swin kernels.
#include "fused_attn.h"
#include "fused_mlp.h"
#include "fused_mlp_x3.h"
#include <atomic>
#include "fused_rowgemm.h"
#include "fused_deembed.h"
#include "launchers.h"

namespace escx {

template <int CP, int TM>
static void launch_mlp(const MlpArgs& a, hipStream_t s) {
    const int waves = (a.M + 16 * TM - 1) / (16 * TM);
    ESCX_LAUNCH((mlp_fused_kernel<CP, TM>), dim3((waves + 3) / 4), dim3(256), 0, s, a);
}

template <int CP, int TM, int NW>
static void launch_mlp_lds(const MlpArgs& a, hipStream_t s) {
    const int rows = 16 * TM * NW;
    const int hs = a.HS > 1 ? a.HS : 1;
#ifdef ESCX_EXPERIMENTAL       // in-launch combine of the hidden split: measured slower (profiles/r4_mlp_combine_ab.txt), tagged builds only
    if constexpr (TM == 1 && (CP == 192 || CP == 384) && (NW == 4 || NW == 8)) {      // the widths that take the hidden split
        if (a.tickets) { ESCX_LAUNCH((mlp_fused_lds_kernel<CP, TM, NW, 0, true>), dim3(((a.M + rows - 1) / rows) * hs), dim3(64 * NW), 0, s, a); return; }
    }
#endif
    MlpArgs b = a; b.tickets = nullptr;
    ESCX_LAUNCH((mlp_fused_lds_kernel<CP, TM, NW>), dim3(((a.M + rows - 1) / rows) * hs), dim3(64 * NW), 0, s, b);
}

// variant: 0 = wave-autonomous; otherwise LDS-staged with (TM, NW) = 1:(1,4) 2:(1,6) 3:(1,8) 4:(2,4) 5:(2,8)
template <int CP>
static int launch_mlp_lds_variant(int variant, const MlpArgs& a, hipStream_t s) {
    switch (variant) {
        case 1: launch_mlp_lds<CP, 1, 4>(a, s); return 0;
        case 3: launch_mlp_lds<CP, 1, 8>(a, s); return 0;
#ifdef ESCX_EXPERIMENTAL       // 6-wave workgroups and two row tiles per wave: measured no better, tagged builds only
        case 2: launch_mlp_lds<CP, 1, 6>(a, s); return 0;
        case 4: if constexpr (CP <= 192) { launch_mlp_lds<CP, 2, 4>(a, s); return 0; } return -1;
        case 5: if constexpr (CP <= 192) { launch_mlp_lds<CP, 2, 8>(a, s); return 0; } return -1;
#endif
        default: return -1;
    }
}

#ifdef ESCX_EXPERIMENTAL
template <int CP, int TM, int NW, int ABL>
static void launch_mlp_abl(const MlpArgs& a, hipStream_t s) {
    const int rows = 16 * TM * NW;
    ESCX_LAUNCH((mlp_fused_lds_kernel<CP, TM, NW, ABL>), dim3((a.M + rows - 1) / rows), dim3(64 * NW), 0, s, a);
}
#endif

static unsigned long long* g_mlp_trace = nullptr;      // debug only (ESCX_MLP_VARIANT=164): 8 x u64 per wave, see fused_mlp.h
void mlp_set_trace(unsigned long long* p) { g_mlp_trace = p; }
unsigned long long* debug_trace_buffer() { return g_mlp_trace; }

void rows_combine(float* dst, const float* src, const float* partial, const float* bias, long long M, int Cp, int n, hipStream_t s) {
    const long long n4 = M * Cp / 4;
    ESCX_LAUNCH(rows_combine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dst, src, partial, bias, M, Cp, n);
}

// *hs: requested hidden split in, split actually used out (> 1: x is untouched, partial[hs][M][Cp] is filled, the caller runs rows_combine)
template <int CP, int NW>
static void launch_mlp_split(const MlpArgs& a, hipStream_t s) {
    const int rows = 16 * NW;
    ESCX_LAUNCH((mlp_fused_lds_kernel<CP, 1, NW, 0, false, true>), dim3((a.M + rows - 1) / rows), dim3(64 * NW), 0, s, a);
}

int mlp_fused(float* x, int M, int C, int Cp, const float* gamma, const float* beta, const float* w1f, const float* b1,
              const float* w2f, const float* b2, const float* wcf, int hiddenP, int variant, int* hs_io, float* partial, hipStream_t s, float* out,
              int* tickets, int n_tickets, bool* combined, const MlpSplit* split) {
    if (split) {        // PatchSplit in the epilogue: the widths whose MLP is not hidden-split (ESC: C = 144, 96, 72), plain 4- / 8-wave variants
        const int hs0 = hs_io ? *hs_io : 1;
        if (hs0 > 1 || out || (variant != 1 && variant != 3) || (split->NT & 1) || !(Cp == 80 || Cp == 96 || Cp == 144)) return ESCX_COMB_UNSUPPORTED;
        if (combined) *combined = false;
        MlpArgs a{x, gamma, beta, reinterpret_cast<const f32x4*>(w1f), b1, reinterpret_cast<const f32x4*>(w2f), b2,
                  reinterpret_cast<const f32x4*>(wcf), M, C, hiddenP / 16, 1e-5f, nullptr, 1, nullptr, nullptr, nullptr,
                  reinterpret_cast<const f32x4*>(split->wf), split->gamma, split->beta, split->out, split->NT, split->H, split->W, split->C2p};
        const bool nw8 = variant == 3;
        switch (Cp) {
            case 80: if (nw8) launch_mlp_split<80, 8>(a, s); else launch_mlp_split<80, 4>(a, s); return 0;
            case 96: if (nw8) launch_mlp_split<96, 8>(a, s); else launch_mlp_split<96, 4>(a, s); return 0;
            case 144: if (nw8) launch_mlp_split<144, 8>(a, s); else launch_mlp_split<144, 4>(a, s); return 0;
        }
        return ESCX_COMB_UNSUPPORTED;
    }
    int hs = hs_io ? *hs_io : 1;
    const bool lds_width = Cp == 48 || Cp == 80 || Cp == 96 || Cp == 144 || Cp == 192 || Cp == 384;
    if (hs > 1 && (variant <= 0 || variant >= 100 || variant > 3 || !lds_width || !partial || (hiddenP / 16) % hs)) hs = 1;
    if (hs_io) *hs_io = hs;
    // combine inside the launch (fused_mlp.h): needs a zeroed arrival counter per row block and slab offsets that fit the 32-bit buffer addressing
    // OPT-IN (ESCX_MLP_FUSED_COMBINE=1).  MEASURED (round 4, B = 36, profiles/r4_mlp_combine_ab.txt): bit-identical to the two-launch form, but the
    // last arriver's serial tail (drain of the write-through stores, ticket, acquire, two slab reads from memory) costs more than the combine launches
    // it removes - mlp C = 192 / 384 +0.44 / +0.36 ms per step against 0.30 ms of combine launches, whole step 16.06 -> 16.45 ms.
    static const bool fuse_combine = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_FUSED_COMBINE"); return e && e[0] == '1'; }();
    const int rows_per_wg = 16 * ((variant == 3) ? 8 : (variant == 2 ? 6 : 4));       // TM = 1 variants only (hs > 1 implies variant 1..3)
    const int n_rb = (M + rows_per_wg - 1) / rows_per_wg;
    const bool in_kernel = hs > 1 && fuse_combine && tickets && n_rb <= n_tickets && (size_t)hs * M * Cp * sizeof(float) < 0xffffffffull && !out &&
                           (Cp == 192 || Cp == 384) && (variant == 1 || variant == 3);
    if (combined) *combined = in_kernel;
    MlpArgs a{x, gamma, beta, reinterpret_cast<const f32x4*>(w1f), b1, reinterpret_cast<const f32x4*>(w2f), b2,
              reinterpret_cast<const f32x4*>(wcf), M, C, hiddenP / 16, 1e-5f, g_mlp_trace, hs, partial, out, in_kernel ? tickets : nullptr,
              nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (out && (hs > 1 || variant >= 100)) return -1;       // a separate output: plain epilogues only (no hidden split, no ablation builds)
#ifdef ESCX_EXPERIMENTAL
    if (variant >= 100) {      // timing-only ablations: variant = 100 + ABL bits
        const int abl = variant - 100;
        if (Cp == 192) {
            switch (abl) {
                case 0: launch_mlp_abl<192, 1, 4, 0>(a, s); return 0;
                case 1: launch_mlp_abl<192, 1, 4, 1>(a, s); return 0;
                case 2: launch_mlp_abl<192, 1, 4, 2>(a, s); return 0;
                case 3: launch_mlp_abl<192, 1, 4, 3>(a, s); return 0;
                case 4: launch_mlp_abl<192, 1, 4, 4>(a, s); return 0;
                case 6: launch_mlp_abl<192, 1, 4, 6>(a, s); return 0;
                case 7: launch_mlp_abl<192, 1, 4, 7>(a, s); return 0;
                case 34: launch_mlp_abl<192, 1, 4, 34>(a, s); return 0;
                case 39: launch_mlp_abl<192, 1, 4, 39>(a, s); return 0;
                case 64: launch_mlp_abl<192, 1, 4, 64>(a, s); return 0;
            }
        }
        if (Cp == 384 && abl == 64) { launch_mlp_abl<384, 1, 4, 64>(a, s); return 0; }
        if (abl == 256) {       // A/B: next stage's DMA issued as one burst after the barrier instead of spread over fc1
            if (Cp == 384) { launch_mlp_abl<384, 1, 4, 256>(a, s); return 0; }
            if (Cp == 192) { launch_mlp_abl<192, 1, 4, 256>(a, s); return 0; }
        }
        if (Cp == 48 && abl == 64) { launch_mlp_abl<48, 1, 8, 64>(a, s); return 0; }
        if (Cp == 48 && abl == 34) { launch_mlp_abl<48, 1, 8, 34>(a, s); return 0; }      // round 5 timing experiment: no stage barrier, no weight DMA (what LDS-resident weights would save)
        if (Cp == 48 && abl == 2) { launch_mlp_abl<48, 1, 8, 2>(a, s); return 0; }
        if (Cp == 48 && abl == 32) { launch_mlp_abl<48, 1, 8, 32>(a, s); return 0; }
        if (Cp == 48 && abl == 0) { launch_mlp_abl<48, 1, 8, 0>(a, s); return 0; }
        variant = 1;
    }
#else
    if (variant >= 100) variant = 1;      // timing-only ablation kernels exist in tagged builds only
#endif
    if (variant > 0) {
        switch (Cp) {
            case 48: return launch_mlp_lds_variant<48>(variant, a, s);
            case 80: return launch_mlp_lds_variant<80>(variant, a, s);
            case 96: return launch_mlp_lds_variant<96>(variant, a, s);
            case 144: return launch_mlp_lds_variant<144>(variant, a, s);
            case 192: return launch_mlp_lds_variant<192>(variant, a, s);
            case 384: return launch_mlp_lds_variant<384>(variant, a, s);
            default: break;     // fall through to the wave-autonomous kernel
        }
    }
    switch (Cp) {
        case 16: launch_mlp<16, 4>(a, s); return 0;
        case 32: launch_mlp<32, 4>(a, s); return 0;
        case 48: launch_mlp<48, 4>(a, s); return 0;
        case 64: launch_mlp<64, 4>(a, s); return 0;
        case 80: launch_mlp<80, 4>(a, s); return 0;
        case 96: launch_mlp<96, 2>(a, s); return 0;
        case 112: launch_mlp<112, 2>(a, s); return 0;
        case 128: launch_mlp<128, 2>(a, s); return 0;
        case 144: launch_mlp<144, 2>(a, s); return 0;
        case 160: launch_mlp<160, 2>(a, s); return 0;
        case 192: launch_mlp<192, 2>(a, s); return 0;
        case 256: launch_mlp<256, 1>(a, s); return 0;
        case 288: launch_mlp<288, 1>(a, s); return 0;
        case 384: launch_mlp<384, 1>(a, s); return 0;
        default: return -1;
    }
}

// ---- fused MLP, fp32 operands split into three bf16 terms (fused_mlp_x3.h) ----
// nt = 3: three bf16 terms per operand (exact split); nt = 2: two fp16 terms + per-matrix power-of-two scales in a 32-byte trailer (fused_mlp_x3.h)
size_t mlp_x3_bytes(int Cp, int hiddenP, int nt) { return (size_t)(hiddenP / 32) * mlp_x3_frags(Cp, nt) * 1024 + 48; }

int mlp_x3_pack(const float* w1, const float* w2, void* image, int Cp, int hiddenP, hipStream_t s, int nt, const float* gamma, const float* beta, const float* b1, int C) {
    if (Cp % 16 || hiddenP % 32 || (nt != 2 && nt != 3)) return -1;
    const int KS = (Cp + 31) / 32, KK = Cp / 16;
    const long long total = (long long)(hiddenP / 32) * (2 * KS + KK) * 64;
    if (nt == 2) {
        unsigned* mx = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(image) + mlp_x3_bytes(Cp, hiddenP, nt) - 48);
        if (hipMemsetAsync(mx, 0, 16, s) != hipSuccess) return -1;
        const long long n = (long long)hiddenP * Cp;
        ESCX_LAUNCH(absmax_bits_kernel, dim3((unsigned)std::min<long long>(256, (n + 255) / 256)), dim3(256), 0, s, w1, n, mx);
        ESCX_LAUNCH(absmax_bits_kernel, dim3((unsigned)std::min<long long>(256, (n + 255) / 256)), dim3(256), 0, s, w2, n, mx + 1);
        ESCX_LAUNCH(rownorm2_max_bits_kernel, dim3((unsigned)((hiddenP + 3) / 4)), dim3(256), 0, s, w1, hiddenP, Cp, Cp, mx + 2);      // range rule: bound of the fc1 outputs
    }
    ESCX_LAUNCH(mlp_x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w1, w2, reinterpret_cast<bf16x8*>(image), Cp, hiddenP, KS, KK, nt, gamma, beta, b1, C);
    return 0;
}

template <int CP, int NW, int NT = 3>
static void launch_mlp_x3(const MlpArgs& a, hipStream_t s) {
    auto kern = mlp_x3_kernel<CP, NW, false, NT>;
    constexpr int lds = 2 * mlp_x3_stage_frags(CP, NT) * 1024;
    if constexpr (lds > 48 * 1024) {            // function attributes are per device: one flag per device
        static std::atomic<unsigned> done{0};
        int dev = 0; (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done.fetch_or(bit, std::memory_order_relaxed); }
    }
    const int hs = a.HS > 1 ? a.HS : 1;
    ESCX_LAUNCH(kern, dim3(((a.M + 16 * NW - 1) / (16 * NW)) * hs), dim3(64 * NW), lds, s, a);
}

#ifdef ESCX_EXPERIMENTAL       // two row tiles per wave: measured slower (fused_mlp_x3.h)
template <int CP, int NW, int TM>
static void launch_mlp_x3_rows(const MlpArgs& a, hipStream_t s) {
    auto kern = mlp_x3_rows_kernel<CP, NW, TM>;
    constexpr int lds = 2 * mlp_x3_stage_frags(CP) * 1024;
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned> done{0};
        int dev = 0; (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done.fetch_or(bit, std::memory_order_relaxed); }
    }
    ESCX_LAUNCH(kern, dim3((a.M + 16 * NW * TM - 1) / (16 * NW * TM)), dim3(64 * NW), lds, s, a);
}
#endif

size_t mlp_x3_split_bytes(int Cp, int Np) { return (size_t)(Np / 16) * 3 * ((Cp / 16 + 1) / 2) * 1024; }
int mlp_x3_split_pack(const float* wf, void* image, int Cp, int Np, hipStream_t s) {
    const int KK = Cp / 16, NT = Np / 16;
    const long long total = (long long)NT * ((KK + 1) / 2) * 64;
    ESCX_LAUNCH(mlp_x3_split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(wf), reinterpret_cast<bf16x8*>(image), NT, KK);
    return 0;
}

template <int CP, int NW, int NT = 3>
static void launch_mlp_x3_split(const MlpArgs& a, hipStream_t s) {
    auto kern = mlp_x3_kernel<CP, NW, true, NT>;
    constexpr int lds = 2 * mlp_x3_stage_frags(CP, NT) * 1024;
    if constexpr (lds > 48 * 1024) {
        static std::atomic<unsigned> done{0};
        int dev = 0; (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done.fetch_or(bit, std::memory_order_relaxed); }
    }
    ESCX_LAUNCH(kern, dim3((a.M + 16 * NW - 1) / (16 * NW)), dim3(64 * NW), lds, s, a);
}

// *hs_io: requested hidden split in, split used out (> 1: x untouched, partial[hs][M][Cp] filled, the caller runs rows_combine) - as mlp_fused
// split: PatchSplit in the epilogue (split->wf = the image of mlp_x3_split_pack); ESCX_COMB_UNSUPPORTED when the width has no such instantiation
int mlp_x3(float* x, int M, int C, int Cp, const float* gamma, const float* beta, const float* b1, const float* b2, const void* image, int hiddenP, int nw, int* hs_io, float* partial,
           hipStream_t s, const MlpSplit* split, int nt, float* out) {
    if (!image || hiddenP % 32 || (nt != 2 && nt != 3)) return -1;
    if (out && (split || (hs_io && *hs_io > 1))) return -1;             // out-of-place: the plain form only (training forward: x1 -> x2)
    if (split) {
        if ((hs_io && *hs_io > 1) || !split->wf || !(Cp == 80 || Cp == 96 || Cp == 144)) return ESCX_COMB_UNSUPPORTED;
        MlpArgs a{};
        a.x = x; a.gamma = gamma; a.beta = beta; a.b1 = b1; a.b2 = b2; a.M = M; a.C = C; a.HT = hiddenP / 16; a.eps = 1e-5f; a.HS = 1; a.x3_w = image;
        a.sp_wf = reinterpret_cast<const f32x4*>(split->wf); a.sp_gamma = split->gamma; a.sp_beta = split->beta; a.sp_out = split->out;
        a.sp_NT = split->NT; a.sp_H = split->H; a.sp_W = split->W; a.sp_C2p = split->C2p;
        switch (Cp) {
#define ESCX_X3S_CASE(CPV) case CPV: if (nt == 2) { if (nw == 8) launch_mlp_x3_split<CPV, 8, 2>(a, s); else launch_mlp_x3_split<CPV, 4, 2>(a, s); } \
                           else { if (nw == 8) launch_mlp_x3_split<CPV, 8>(a, s); else launch_mlp_x3_split<CPV, 4>(a, s); } return 0;
            ESCX_X3S_CASE(80) ESCX_X3S_CASE(96) ESCX_X3S_CASE(144)
#undef ESCX_X3S_CASE
        }
        return ESCX_COMB_UNSUPPORTED;
    }
    int hs = hs_io ? *hs_io : 1;
    if (hs > 1 && (!partial || (hiddenP / 32) % hs)) hs = 1;
    if (hs_io) *hs_io = hs;
    MlpArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.b1 = b1; a.b2 = b2; a.M = M; a.C = C; a.HT = hiddenP / 16; a.eps = 1e-5f; a.HS = hs; a.partial = partial; a.x3_w = image; a.out = out;
#ifdef ESCX_EXPERIMENTAL
    // two row tiles per wave (mlp_x3_rows_kernel, bit-identical, measured slower) for the narrow maps without a hidden split: ESCX_MLP_X3_TM=2
    static const int tm_env = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_X3_TM"); return e ? atoi(e) : 1; }();
    if (hs == 1 && tm_env == 2 && Cp <= 96) {
        switch (Cp) {
            case 48: if (nw == 8) launch_mlp_x3_rows<48, 8, 2>(a, s); else launch_mlp_x3_rows<48, 4, 2>(a, s); return 0;
            case 80: if (nw == 8) launch_mlp_x3_rows<80, 8, 2>(a, s); else launch_mlp_x3_rows<80, 4, 2>(a, s); return 0;
            case 96: if (nw == 8) launch_mlp_x3_rows<96, 8, 2>(a, s); else launch_mlp_x3_rows<96, 4, 2>(a, s); return 0;
        }
    }
#endif
#define ESCX_X3_CASE(CPV) case CPV: if (nt == 2) { if (nw == 8) launch_mlp_x3<CPV, 8, 2>(a, s); else launch_mlp_x3<CPV, 4, 2>(a, s); } \
                          else { if (nw == 8) launch_mlp_x3<CPV, 8>(a, s); else launch_mlp_x3<CPV, 4>(a, s); } return 0;
    switch (Cp) {
        ESCX_X3_CASE(48) ESCX_X3_CASE(80) ESCX_X3_CASE(96) ESCX_X3_CASE(144) ESCX_X3_CASE(192) ESCX_X3_CASE(384)
        default: return -1;
    }
#undef ESCX_X3_CASE
}

// ---- fused LN + linear for PatchMerge / PatchSplit ----
#ifdef ESCX_EXPERIMENTAL       // weight-stationary / shared-rows forms: faster alone, slower in the two-stream step (profiles/r4_rowgemm_ab.txt)
// Weight-stationary persistent form (fused_rowgemm.h).  Chunking over output tiles never changes an output element's arithmetic, so it
// may depend on the batch: the LDS budget bounds a chunk from above, and small grids (few row groups) take more, smaller chunks so that
// every SIMD gets a wave.
constexpr int rowgemm_ws_wps(int regs) { return regs <= 64 ? 8 : (regs <= 80 ? 6 : (regs <= 96 ? 5 : (regs <= 128 ? 4 : (regs <= 168 ? 3 : (regs <= 256 ? 2 : 1))))); }

template <int KP, int SEGS, int NW, bool PF>
static void launch_rowgemm_ws(const RowGemmArgs& a, hipStream_t s) {
    constexpr int KK = KP / 16;
    constexpr int TM = KP <= 192 ? 2 : 1;                       // as rowgemm_fused_kernel: the accumulator split (and with it every sum) is unchanged
    constexpr int REGS = TM * KK * 4 * (PF ? 2 : 1) + 100;      // operand tile(s) + what hipcc measurably needs around them (ring, accumulators, addresses, LayerNorm temporaries)
    constexpr int WPS0 = rowgemm_ws_wps(REGS);
    constexpr int WPS = (WPS0 * 4 < NW) ? (NW / 4) : WPS0;        // one workgroup must fit a CU
    auto kern = rowgemm_ws_kernel<KP, SEGS, TM, NW, WPS, PF>;
    static const int lds_cap = [] { const char* e = ESCX_TUNE_ENV("ESCX_RG_LDS_KB"); return (e && e[0] ? atoi(e) : 144) * 1024; }();
    const int tile_bytes = KK * 1024;
    const int max_tiles = std::max(1, lds_cap / tile_bytes);
    const int n_groups = (a.M + 16 * TM - 1) / (16 * TM);
    int chunks = (a.NT + max_tiles - 1) / max_tiles;
    while (chunks < a.NT && (long long)n_groups * chunks < 1024) ++chunks;         // small grids: a wave for every SIMD
    RowGemmArgs b = a;
    b.nt_chunk = (a.NT + chunks - 1) / chunks;
    chunks = (a.NT + b.nt_chunk - 1) / b.nt_chunk;
    const int lds = b.nt_chunk * tile_bytes;
    const int by_lds = std::max(1, (160 * 1024) / lds), by_waves = std::max(1, std::min(32 / NW, WPS * 4 / NW));
    const int wg_per_cu = std::min(by_lds, by_waves);
    int gx = std::max(1, 256 * wg_per_cu / chunks);
    gx = std::min(gx, (n_groups + NW - 1) / NW);
    {   // every wave the same number of row groups: with `it` passes over gx * NW waves, shrink gx until the last pass is (nearly) full
        const int it = (n_groups + gx * NW - 1) / (gx * NW);
        gx = std::max(1, (n_groups + it * NW - 1) / (it * NW));
    }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      // per launch: function attributes are per device
    ESCX_LAUNCH(kern, dim3(gx, chunks), dim3(64 * NW), lds, s, b);
}

template <int KP, int SEGS>
static bool launch_rowgemm_ws_variant(const RowGemmArgs& a, hipStream_t s) {
    // 0 (default): streaming kernel; 2..5: weight-stationary form with (NW, PF) = (4, no), (4, yes), (8, no), (8, yes).  MEASURED (round 4, B = 36,
    // profiles/r4_rowgemm_ab.txt): alone on the GPU the new forms are 5-25 % faster per launch, but in the product's two-stream execution the step gets
    // 0.1-0.2 ms SLOWER (the co-running MLP / attention launches stretch by more than these kernels shrink), so they stay opt-in.
    static const int mode = [] { const char* e = ESCX_TUNE_ENV("ESCX_ROWGEMM_WS"); return e && e[0] ? atoi(e) : 0; }();
    if (mode == 0) return false;
    constexpr int KK = KP / 16;
    switch (mode) {
        case 2: launch_rowgemm_ws<KP, SEGS, 4, false>(a, s); return true;
        case 3: launch_rowgemm_ws<KP, SEGS, 4, true>(a, s); return true;
        case 4: launch_rowgemm_ws<KP, SEGS, 8, false>(a, s); return true;
        case 5: launch_rowgemm_ws<KP, SEGS, 8, true>(a, s); return true;
        default: return false;
    }
}

// Shared-rows form for the deep scales (fused_rowgemm.h (B)): 3 row tiles per workgroup, 3 output tiles per wave.
template <int KP, int SEGS, int NW>
static void launch_rowgemm_xs(const RowGemmArgs& a, hipStream_t s) {
    constexpr int KK = KP / 16, R = 3, NTW = 3;
    constexpr int WPS0 = KP <= 192 ? 3 : 2;
    constexpr int WPS = (WPS0 * 4 < NW) ? (NW / 4) : WPS0;
    auto kern = rowgemm_xs_kernel<KP, SEGS, R, NW, NTW, WPS>;
    const int lds = R * KK * 1024;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);             // per launch: function attributes are per device
    ESCX_LAUNCH(kern, dim3((a.M + 16 * R - 1) / (16 * R)), dim3(64 * NW), lds, s, a);
}

template <int KP, int SEGS>
static bool launch_rowgemm_xs_variant(const RowGemmArgs& a, hipStream_t s) {
    if constexpr (KP >= 144) {
        static const bool on = [] { const char* e = ESCX_TUNE_ENV("ESCX_ROWGEMM_XS"); return e && e[0] == '1'; }();      // opt-in, see launch_rowgemm_ws_variant
        if (!on || a.NT * (KP / 16) < 96) return false;         // matrices under ~96 KB stay with the weight-stationary form
        switch ((a.NT + 2) / 3) {
            case 3: launch_rowgemm_xs<KP, SEGS, 3>(a, s); return true;
            case 4: launch_rowgemm_xs<KP, SEGS, 4>(a, s); return true;
            case 6: launch_rowgemm_xs<KP, SEGS, 6>(a, s); return true;
            case 8: launch_rowgemm_xs<KP, SEGS, 8>(a, s); return true;
            default: return false;
        }
    }
    return false;
}

#endif

template <int KP, int SEGS>
static int launch_rowgemm(const RowGemmArgs& a, hipStream_t s) {
    if (a.x3_wf && a.comb_n == 0) {           // split-operand form (fused_rowgemm.h rowgemm_x3_kernel): the widths of the ESC scale changes
        if constexpr (KP == 96 || KP == 160 || KP == 192 || KP == 288 || KP == 384 || KP == 144 || KP == 80) {
            constexpr int KSx = (KP + 31) / 32, TFx = 3 * KSx, UTx = 36 / TFx >= 4 ? 4 : (36 / TFx >= 2 ? 2 : 1);
            constexpr int TMx = KP <= 96 ? 2 : 1, NWx = 4;           // one row tile per wave above K = 96: the three-term operand costs 1.5x the registers of the fp32 one
            auto kern = rowgemm_x3_kernel<KP, SEGS, TMx, NWx, UTx>;
            auto kern2 = rowgemm_x3_kernel<KP, SEGS, TMx, NWx, UTx, 2>;
            constexpr int lds = 2 * UTx * TFx * 1024;
            if constexpr (lds > 48 * 1024) {
                static std::atomic<unsigned> done{0};
                int dev = 0; (void)hipGetDevice(&dev);
                const unsigned bit = 1u << (dev & 31);
                if (!(done.load(std::memory_order_relaxed) & bit)) {
                    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    (void)hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    done.fetch_or(bit, std::memory_order_relaxed);
                }
            }
            if (a.x3_scale) ESCX_LAUNCH(kern2, dim3((a.M + 16 * TMx * NWx - 1) / (16 * TMx * NWx), 1), dim3(64 * NWx), lds, s, a);
            else ESCX_LAUNCH(kern, dim3((a.M + 16 * TMx * NWx - 1) / (16 * TMx * NWx), 1), dim3(64 * NWx), lds, s, a);
            return 0;
        }
    }
    if (a.comb_n > 0) {
#ifdef ESCX_EXPERIMENTAL       // combine on load: measured slower (profiles/r4_mlp_combine_ab.txt)
        constexpr bool HAS_COMB = (KP == 384) || (KP == 192 && SEGS == 1);      // the scale changes that follow a hidden-split MLP (C = 192 / 384)
        if constexpr (HAS_COMB) {
            constexpr int KK = KP / 16;
            constexpr int UT = KK <= 6 ? 4 : (KK <= 12 ? 2 : 1);
            constexpr int TM = KP <= 192 ? 2 : 1;
            constexpr int NW = 4;
            RowGemmArgs b = a; b.nt_chunk = a.NT;
            ESCX_LAUNCH((rowgemm_fused_kernel<KP, SEGS, TM, NW, UT, true>), dim3((a.M + 16 * TM * NW - 1) / (16 * TM * NW), 1), dim3(64 * NW), 0, s, b);
            return 0;
        }
#endif
        return ESCX_COMB_UNSUPPORTED;
    }
#ifdef ESCX_EXPERIMENTAL
    if (launch_rowgemm_xs_variant<KP, SEGS>(a, s)) return 0;
    if (launch_rowgemm_ws_variant<KP, SEGS>(a, s)) return 0;
#endif
    constexpr int KK = KP / 16;
    constexpr int UT = KK <= 6 ? 4 : (KK <= 12 ? 2 : 1);
    constexpr int TM = KP <= 192 ? 2 : 1;
    constexpr int NW = 4;
    const int rows = 16 * TM * NW;
    // Output-column chunks (fused_rowgemm.h), bit-identical for every chunking.  MEASURED (round 3, B = 36): targets of 3072 / 6144 waves make the
    // merge + split kernels 3 % / 8 % SLOWER alone (1.333 -> 1.379 / 1.452 ms per step) and the step 2 - 2.6 % slower: these kernels are not short of
    // waves, the re-done gather + LayerNorm costs more than the extra occupancy returns.  Off by default (ESCX_ROWGEMM_WAVES = target to try it).
    static const int target = [] { const char* e = ESCX_TUNE_ENV("ESCX_ROWGEMM_WAVES"); return e ? atoi(e) : 0; }();
    RowGemmArgs b = a;
    const int waves = (a.M + 16 * TM - 1) / (16 * TM);
    int chunks = target > 0 ? (target + waves - 1) / waves : 1;
    chunks = std::max(1, std::min(chunks, a.NT / std::max(2 * UT, 2)));
    b.nt_chunk = (a.NT + chunks - 1) / chunks;
    b.nt_chunk = (b.nt_chunk + UT - 1) / UT * UT;                 // whole stages
    chunks = (a.NT + b.nt_chunk - 1) / b.nt_chunk;
    ESCX_LAUNCH((rowgemm_fused_kernel<KP, SEGS, TM, NW, UT>), dim3((a.M + rows - 1) / rows, chunks), dim3(64 * NW), 0, s, b);
    return 0;
}

// 32-byte trailer: the two-term (nt = 2) form keeps max |w| and its power-of-two scales there (fused_attn.h attn_x3_pack_kernel)
size_t rowgemm_x3_bytes(int KP, int Np) { return (size_t)(Np / 16) * 3 * ((KP + 31) / 32) * 1024 + 32; }
int rowgemm_x3_pack(const float* wf, void* image, int KP, int Np, hipStream_t s, int nt, const float* gamma, const float* beta, int C) {
    const int KK = KP / 16, KS = (KP + 31) / 32, NT = Np / 16;
    const long long total = (long long)NT * (KS > KK ? KS : KK) * 64;
    if (nt == 2) {
        unsigned* mx = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(image) + rowgemm_x3_bytes(KP, Np) - 32);
        (void)hipMemsetAsync(mx, 0, 16, s);
        const long long n = (long long)NT * KK * 64 * 4;
        ESCX_LAUNCH(absmax_bits_kernel, dim3((unsigned)std::min<long long>(256, (n + 255) / 256)), dim3(256), 0, s, wf, n, mx);
    }
    ESCX_LAUNCH(attn_x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(wf), reinterpret_cast<bf16x8*>(image),
                       NT, 1, KK, KS, 3 * KS, 0u, nt == 2 ? 2 : 3, gamma, beta, KP, C, (const float*)nullptr, 0);
    return 0;
}

int rowgemm_fused(int segs, const float* x, float* out, const float* gamma, const float* beta, const float* wf, const int* map, int M,
                  int rows_per_clip, int src_rows_per_clip, int C, int Cp, int Np, int split, int H, int W, int C2p, hipStream_t s,
                  const CombineOnLoad* comb, const void* x3_wf, int x3_nt) {
    const int KP = segs * Cp;
    RowGemmArgs a{x, out, gamma, beta, reinterpret_cast<const f32x4*>(wf), map, M, rows_per_clip, src_rows_per_clip, C, Cp, Np / 16,
                  split, H, W, C2p, 1e-5f, Np / 16, comb ? comb->partial : nullptr, comb ? comb->bias : nullptr, comb ? comb->stride : 0, comb ? comb->n : 0,
                  comb ? nullptr : x3_wf,
                  (!comb && x3_wf && x3_nt == 2) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(x3_wf) + rowgemm_x3_bytes(KP, Np) - 16) : nullptr};
    if (segs == 1) {
        switch (KP) {
            case 16: return launch_rowgemm<16, 1>(a, s);
            case 48: return launch_rowgemm<48, 1>(a, s);
            case 80: return launch_rowgemm<80, 1>(a, s);
            case 96: return launch_rowgemm<96, 1>(a, s);
            case 144: return launch_rowgemm<144, 1>(a, s);
            case 192: return launch_rowgemm<192, 1>(a, s);
            case 384: return launch_rowgemm<384, 1>(a, s);
            default: return -1;
        }
    }
    switch (KP) {
        case 32: return launch_rowgemm<32, 2>(a, s);
        case 96: return launch_rowgemm<96, 2>(a, s);
        case 160: return launch_rowgemm<160, 2>(a, s);
        case 192: return launch_rowgemm<192, 2>(a, s);
        case 288: return launch_rowgemm<288, 2>(a, s);
        case 384: return launch_rowgemm<384, 2>(a, s);
        default: return -1;
    }
}

// ---- halo-tiled composed de-embedding -----------------------------------------------------------
// the two-term fp16 weight stream of deembed7_x2_kernel (fused_deembed.h), from the fp32 fragment stream; 0 bytes: no such instantiation for this width
size_t deembed7_x2_image_bytes(int Cp) { return Cp == 48 ? deembed7_x2_bytes(Cp) : 0; }
int deembed7_x2_pack(const float* wfrag, void* image, int Cp, hipStream_t s) {
    if (Cp != 48) return -1;
    const int nsteps = de2_steps(Cp);
    unsigned* mx = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(image) + deembed7_x2_bytes(Cp) - 32);
    if (hipMemsetAsync(mx, 0, 32, s) != hipSuccess) return -1;
    const long long n = (long long)49 * (Cp / 16) * 64 * 4;
    ESCX_LAUNCH(absmax_bits_kernel, dim3((unsigned)std::min<long long>(256, (n + 255) / 256)), dim3(256), 0, s, wfrag, n, mx);
    ESCX_LAUNCH(deembed7_x2_pack_kernel, dim3((nsteps * 64 + 255) / 256), dim3(256), 0, s, reinterpret_cast<const f32x4*>(wfrag), reinterpret_cast<bf16x8*>(image), Cp, nsteps);
    return 0;
}

int deembed7_fused(const float* tok, int B, int H, int W, int Cp, const float* wfrag, const float* bias, float* out, int pf, int pt,
                   int in_dim, int Fp, hipStream_t s, const void* x2_image) {
    DeembedArgs a{tok, reinterpret_cast<const f32x4*>(wfrag), bias, out, B, H, W, pf, pt, in_dim, Fp, in_dim * pf * pt};
    if (a.n_out > 16) return -1;
    const int grid = B * ((H + 7) / 8) * ((W + 31) / 32);
    if (x2_image && Cp == 48) {                 // two fp16 terms, contraction flattened over the 49 taps (the spectrum feeds the ISTFT only)
        auto kern = deembed7_x2_kernel<48>;
        constexpr int lds = 2 * 14 * 38 * (48 + 8) * 2 + 2 * 8 * 2 * 1024;
        static std::atomic<unsigned> done{0};
        int dev = 0; (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done.fetch_or(bit, std::memory_order_relaxed); }
        ESCX_LAUNCH(kern, dim3(grid), dim3(512), lds, s, a, reinterpret_cast<const bf16x8*>(x2_image));
        return 0;
    }
    switch (Cp) {               // LDS: (8+6) x (32+6) pixels x (Cp+4) dwords + two 7-tap weight stages <= 160 KiB
        case 16: ESCX_LAUNCH(deembed7_kernel<16>, dim3(grid), dim3(512), 0, s, a); return 0;
        case 32: ESCX_LAUNCH(deembed7_kernel<32>, dim3(grid), dim3(512), 0, s, a); return 0;
        case 48: ESCX_LAUNCH(deembed7_kernel<48>, dim3(grid), dim3(512), 0, s, a); return 0;
        default: return -1;
    }
}

// ---- fused window attention --------------------------------------------------------------------
template <int CP, int MODE, int NW>
static int launch_attn(const AttnArgs& a, hipStream_t s) {
    constexpr int UT = CP <= 96 ? 4 : (CP <= 192 ? 2 : 1);
    constexpr int TMW = attn_windows_per_wave(CP);
    const int per_block = TMW * NW;
    const int gs = a.GS > 1 ? a.GS : 1;
    if (a.tape_qkv) {               // training forward (TAPE instantiations for the memory-bound widths; the deep scales keep their GEMMs)
        if constexpr (CP == 16 || CP == 48 || CP == 80 || CP == 96) {      // C = 144 / 192 measured: no gain over their GEMM sequence (62.5 vs 62.3-62.6 ms per step)
            if (gs == 1 && a.comb_n == 0) {
                ESCX_LAUNCH((attn_fused_kernel<CP, MODE, UT, TMW, NW, false, true>), dim3((a.n_windows + per_block - 1) / per_block), dim3(64 * NW), 0, s, a);
                return 0;
            }
        }
        return ESCX_COMB_UNSUPPORTED;
    }
    if (a.comb_n > 0) {
#ifdef ESCX_EXPERIMENTAL
        if constexpr (CP == 192 && MODE == 1) {        // the C = 192 blocks that follow a hidden-split MLP (ESC-Base / Large: 24 heads of 8)
            if (gs == 1) { ESCX_LAUNCH((attn_fused_kernel<CP, MODE, UT, TMW, NW, true>), dim3((a.n_windows + per_block - 1) / per_block), dim3(64 * NW), 0, s, a); return 0; }
        }
#endif
        return ESCX_COMB_UNSUPPORTED;
    }
    // split-operand (3 x bf16) Q / K / V projections: the (width, head mapping) pairs of the ESC configurations
    if constexpr ((CP == 48 && MODE == 0) || (CP == 80 && (MODE == 0 || MODE == 2)) || (CP == 96 && MODE != 2) || (CP == 144 && MODE != 2) || (CP == 192 && MODE == 1)) {
#ifdef ESCX_EXPERIMENTAL       // pair-wise split output projection: measured no faster (profiles/r5_attn_ab.txt), tagged builds only
        if constexpr (MODE != 2 && CP != 48) {
        if (a.x3_wf && a.x3_pairs) {        // pair-order stream: the output projection in split form too
            auto kern = attn_fused_kernel<CP, MODE, UT, TMW, NW, false, false, true, true>;
            constexpr int lds = 2 * UT * attn_x3_tf(CP) * 1024;
            if constexpr (lds > 48 * 1024) {
                static std::atomic<unsigned> done{0};
                int dev = 0; (void)hipGetDevice(&dev);
                const unsigned bit = 1u << (dev & 31);
                if (!(done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done.fetch_or(bit, std::memory_order_relaxed); }
            }
            ESCX_LAUNCH(kern, dim3(((a.n_windows + per_block - 1) / per_block) * gs), dim3(64 * NW), lds, s, a);
            return 0;
        }
        }
#endif
        if (a.x3_wf && !a.x3_pairs) {
            auto kern = attn_fused_kernel<CP, MODE, UT, TMW, NW, false, false, true>;
            auto kern2 = attn_fused_kernel<CP, MODE, UT, TMW, NW, false, false, true, false, 2>;        // two fp16 terms (split_terms.h)
            constexpr int lds = 2 * UT * attn_x3_tf(CP) * 1024;
            if constexpr (lds > 48 * 1024) {
                static std::atomic<unsigned> done{0};
                int dev = 0; (void)hipGetDevice(&dev);
                const unsigned bit = 1u << (dev & 31);
                if (!(done.load(std::memory_order_relaxed) & bit)) {
                    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    (void)hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    done.fetch_or(bit, std::memory_order_relaxed);
                }
            }
            if (a.x3_scale) ESCX_LAUNCH(kern2, dim3(((a.n_windows + per_block - 1) / per_block) * gs), dim3(64 * NW), lds, s, a);
            else ESCX_LAUNCH(kern, dim3(((a.n_windows + per_block - 1) / per_block) * gs), dim3(64 * NW), lds, s, a);
            return 0;
        }
    }
    ESCX_LAUNCH((attn_fused_kernel<CP, MODE, UT, TMW, NW>), dim3(((a.n_windows + per_block - 1) / per_block) * gs), dim3(64 * NW), 0, s, a);
    return 0;
}

template <int CP>
static int launch_attn_cp(int mode, int nw, const AttnArgs& a, hipStream_t s) {
    if (nw == 8) {
        switch (mode) {
            case 0: return launch_attn<CP, 0, 8>(a, s);
            case 1: return launch_attn<CP, 1, 8>(a, s);
            case 2: return launch_attn<CP, 2, 8>(a, s);
        }
    } else {
        switch (mode) {
            case 0: return launch_attn<CP, 0, 4>(a, s);
            case 1: return launch_attn<CP, 1, 4>(a, s);
            case 2: return launch_attn<CP, 2, 4>(a, s);
        }
    }
    return -1;
}

template <int CP, int NW>
static int launch_attn_packed(const AttnArgs& a, hipStream_t s) {
    constexpr int UT = CP <= 96 ? 4 : (CP <= 192 ? 2 : 1);
    const int pairs = (a.n_windows + 1) / 2;
    const int gs = a.GS > 1 ? a.GS : 1;
    if (a.comb_n > 0) {
#ifdef ESCX_EXPERIMENTAL
        if constexpr (CP == 384 && NW == 4) {
            if (gs == 1) { ESCX_LAUNCH((attn_packed_kernel<CP, UT, NW, true>), dim3((pairs + NW - 1) / NW), dim3(64 * NW), 0, s, a); return 0; }
        }
#endif
        return ESCX_COMB_UNSUPPORTED;
    }
    if constexpr (CP == 384 && NW == 4) {        // split-operand Q / K / V projections (ESC's bottom scale)
#ifdef ESCX_EXPERIMENTAL
        if (a.x3_wf && a.x3_pairs) {
            auto kern = attn_packed_kernel<CP, UT, NW, false, true, true>;
            constexpr int lds = 2 * UT * attn_x3_tf(CP) * 1024;
            static std::atomic<unsigned> done{0};
            int dev = 0; (void)hipGetDevice(&dev);
            const unsigned bit = 1u << (dev & 31);
            if (!(done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done.fetch_or(bit, std::memory_order_relaxed); }
            ESCX_LAUNCH(kern, dim3(((pairs + NW - 1) / NW) * gs), dim3(64 * NW), lds, s, a);
            return 0;
        }
#endif
        if (a.x3_wf && !a.x3_pairs) {
            auto kern = attn_packed_kernel<CP, UT, NW, false, true>;
            auto kern2 = attn_packed_kernel<CP, UT, NW, false, true, false, 2>;
            constexpr int lds = 2 * UT * attn_x3_tf(CP) * 1024;
            static std::atomic<unsigned> done{0};
            int dev = 0; (void)hipGetDevice(&dev);
            const unsigned bit = 1u << (dev & 31);
            if (!(done.load(std::memory_order_relaxed) & bit)) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                (void)hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                done.fetch_or(bit, std::memory_order_relaxed);
            }
            if (a.x3_scale) ESCX_LAUNCH(kern2, dim3(((pairs + NW - 1) / NW) * gs), dim3(64 * NW), lds, s, a);
            else ESCX_LAUNCH(kern, dim3(((pairs + NW - 1) / NW) * gs), dim3(64 * NW), lds, s, a);
            return 0;
        }
    }
    ESCX_LAUNCH((attn_packed_kernel<CP, UT, NW>), dim3(((pairs + NW - 1) / NW) * gs), dim3(64 * NW), 0, s, a);
    return 0;
}

size_t attn_x3_bytes(int Cp, int mode, int n_groups) { return (size_t)n_groups * (mode == 2 ? 8 : 4) * attn_x3_tf(Cp) * 1024 + 48; }      // + trailer of the two-term form (three maxima, scales: fused_attn.h attn_x3_pack_kernel)

// pairs == 1: pair-order stream (mode 0 / 1, even group count): [Q0 K0 V0 Q1 K1 V1 P_lo P_hi] per two head groups, projection split as well
// pairs == 2: the two-term fp16 form of the plain stream (split_terms.h): weights scaled by the power of two of the block's max |w|, scales in the trailer
int attn_x3_pack(const float* waf, void* image, int Cp, int mode, int n_groups, hipStream_t s, int pairs, const float* gamma, const float* beta, int C, const float* bqkv) {
    const int KK = Cp / 16, KS = attn_x3_ks(Cp), TF = attn_x3_tf(Cp), TPG = mode == 2 ? 8 : 4;
    if (pairs == 1) {
        if (mode == 2 || (n_groups & 1)) return -1;
        (void)hipMemsetAsync(image, 0, attn_x3_bytes(Cp, mode, n_groups), s);
        const int H = (KK + 1) / 2;
        const long long tot = (long long)(n_groups / 2) * 8 * (KS > H ? KS : H) * 64;
        ESCX_LAUNCH(attn_x3p_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(waf), reinterpret_cast<bf16x8*>(image),
                           n_groups / 2, KK, KS, TF);
        return 0;
    }
    const unsigned proj_mask = mode == 2 ? ((1u << 5) | (1u << 7)) : (1u << 3);        // stream order [Q, K, V, P] / [Q_lo, K_lo, Q_hi, K_hi, V_lo, P_lo, V_hi, P_hi]
    const int n_tiles = n_groups * TPG;
    (void)hipMemsetAsync(image, 0, attn_x3_bytes(Cp, mode, n_groups), s);
    const long long total = (long long)n_tiles * (KS > KK ? KS : KK) * 64;
    if (pairs == 2) {
        unsigned* mx = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(image) + attn_x3_bytes(Cp, mode, n_groups) - 48);
        const long long n = (long long)n_tiles * KK * 64 * 4;
        const unsigned all = (1u << TPG) - 1, vmask = mode == 2 ? ((1u << 4) | (1u << 6)) : (1u << 2);
        const dim3 grid((unsigned)std::min<long long>(256, (n + 255) / 256));
        ESCX_LAUNCH(absmax_tiles_bits_kernel, grid, dim3(256), 0, s, waf, n, KK * 256, TPG, all & ~proj_mask, mx);
        ESCX_LAUNCH(absmax_tiles_bits_kernel, grid, dim3(256), 0, s, waf, n, KK * 256, TPG, vmask, mx + 1);
        ESCX_LAUNCH(absmax_tiles_bits_kernel, grid, dim3(256), 0, s, waf, n, KK * 256, TPG, proj_mask, mx + 2);
    }
    ESCX_LAUNCH(attn_x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(waf), reinterpret_cast<bf16x8*>(image),
                       n_tiles, TPG, KK, KS, TF, proj_mask, pairs == 2 ? 2 : 3, gamma, beta, Cp, C, bqkv, n_groups * (mode == 2 ? 6 : 3) * 16);
    return 0;
}

int attn_fused(const float* src, float* dst, int Cp, int C, int mode, int n_groups, const float* gamma, const float* beta,
               const float* wf, const float* bqkv, const float* bias_tab, const float* bproj, const int* map, int slots, int tokens,
               int n_windows, int nWh, int nWw, int shifted, float scale, int nw, int* gs_io, float* partial, int rows, hipStream_t s,
               const CombineOnLoad* comb, const AttnTape* tape, const void* x3_wf, int x3_pairs) {
    int gs = gs_io ? *gs_io : 1;        // head-group split: same in/out convention as mlp_fused
    if (gs > 1 && (!partial || n_groups % gs)) gs = 1;
    if (gs_io) *gs_io = gs;
    AttnArgs a{src, dst, gamma, beta, reinterpret_cast<const f32x4*>(wf), bqkv, bias_tab, bproj, map, slots, tokens, n_windows,
               nWh, nWw, shifted, C, n_groups, scale, 1e-5f, gs, partial, rows, g_mlp_trace,
               comb ? comb->partial : nullptr, comb ? comb->bias : nullptr, comb ? comb->stride : 0, comb ? comb->n : 0,
               tape ? tape->xn : nullptr, tape ? tape->qkv : nullptr, tape ? tape->o : nullptr, tape ? tape->ldq : 0, tape ? tape->ldo : 0,
               tape ? tape->hdp : 0, tape ? tape->nH : 0, (comb || tape) ? nullptr : x3_wf, x3_pairs == 1 ? 1 : 0,
               (!comb && !tape && x3_wf && x3_pairs == 2) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(x3_wf) + attn_x3_bytes(Cp, mode, n_groups) - 32) : nullptr};
    if (tape && nw < 0) return ESCX_COMB_UNSUPPORTED;      // the packed H = 2 form has no tape stores
    // H == 2 scale with no padding along W: two half-real windows share one tile (nw < 0 encodes "packing allowed", |nw| waves)
    if (nw < 0) {
        nw = -nw;
        if (mode == 0) {
#define ESCX_PACK(CPV) case CPV: return nw == 8 ? launch_attn_packed<CPV, 8>(a, s) : launch_attn_packed<CPV, 4>(a, s);
            switch (Cp) { ESCX_PACK(64) ESCX_PACK(96) ESCX_PACK(128) ESCX_PACK(192) ESCX_PACK(256) ESCX_PACK(384) default: break; }
#undef ESCX_PACK
        }
    }
    switch (Cp) {
        case 16: return launch_attn_cp<16>(mode, nw, a, s);
        case 32: return launch_attn_cp<32>(mode, nw, a, s);
        case 48: return launch_attn_cp<48>(mode, nw, a, s);
        case 64: return launch_attn_cp<64>(mode, nw, a, s);
        case 80: return launch_attn_cp<80>(mode, nw, a, s);
        case 96: return launch_attn_cp<96>(mode, nw, a, s);
        case 128: return launch_attn_cp<128>(mode, nw, a, s);
        case 144: return launch_attn_cp<144>(mode, nw, a, s);
        case 192: return launch_attn_cp<192>(mode, nw, a, s);
        case 256: return launch_attn_cp<256>(mode, nw, a, s);
        case 384: return launch_attn_cp<384>(mode, nw, a, s);
        default: return -1;
    }
}

}  // namespace escx
