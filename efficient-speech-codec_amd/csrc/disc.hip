// DAC discriminator + GAN losses of the adversarial training step on MI355X (BASELINE configs[4]).
// Reference: esc/models/discriminator.py:31-221 (MPD, MRD, Discriminator; MSD is unused by every ESC config: rates = []),
// esc/modules/loss/gan_loss.py:5-51, scripts/trainer_adv.py:61-107.  fp32 like the reference.
//
// One handle = one Discriminator.  Parameters live in a flat fp32 device buffer owned by the caller (reference state_dict order of the
// trainable entries: per convolution bias, weight_g, weight_v); the weight-normalised, packed GEMM operands are rebuilt from it at the start of
// every forward (108 small kernels; the weights change every step anyway).  Feature maps are channels-last [B][D0][D1][Cp] buffers owned by
// the caller (they ARE the autograd tape); backward takes the gradients of all feature maps and returns d loss / d parameters and / or
// d loss / d waveform.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <string>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "escx_internal.h"
#include "launchers.h"
#include "train_kernels.h"
#include "disc_kernels.h"
#include "gemm_bf16.h"
#include "conv32_halo.h"

using namespace escx;

namespace {

inline unsigned blk(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

struct DConv {
    std::string prefix;
    int Cin, Cout, CinP, CoutP, CinR, T0, T1, s0, s1, p0, p1, act;
    int Kf, Kt;
    size_t off_b = 0, off_g = 0, off_v = 0;
    float *Wf = nullptr, *Wt = nullptr, *bias = nullptr, *scale = nullptr;
    float* Wp = nullptr; size_t wp_floats = 0;      // strided layers: per-residue-class compact dX weights (disc_kernels.h ConvTSP)
};
struct DSub {
    int kind, arg;                       // 0: MPD(period), 1: MRD(window length)
    std::vector<DConv> convs;            // MPD: 5 + post; MRD: nb * 5 (band-major) + post
    std::vector<std::pair<int, int>> bands;
    float *D = nullptr, *DT = nullptr;   // MRD: windowed DFT matrix [2Fq][w] and its transpose
    int Fq = 0;
};
struct FmapShape { int sub, C, Cp, D0, D1, P1, off1; long long base; };     // off1/P1: slice of a concatenated buffer (MRD band tops); base: first fmap of the slice's buffer

}  // namespace

struct escx_disc_s {
    int device = 0, sample_rate = 16000;
    std::vector<DSub> subs;
    std::vector<std::string> keys; std::vector<size_t> offs, numels;
    size_t total = 0;
    float* wbuf = nullptr; size_t wfloats = 0;
    float* scratch = nullptr; size_t scratch_bytes = 0;
    const float* packed_ptr = nullptr; long long packed_version = -1;      // which (buffer, version) the packed weights were derived from
    int precision = 0;                   // escx_disc_set_precision: 1 = bf16 MFMA for the convolutions, 2 = three-term split operands (fp32-grade) for the wide ones
    __bf16* wbuf16 = nullptr; const float* w16_ptr = nullptr; long long w16_version = -2; int w16_mode = 0;       // bf16 image of wbuf (precision 1), and what it was derived from
    hipStream_t aux[3] = {nullptr, nullptr, nullptr}; hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};    // extra streams: the sub-discriminators are independent of each other
};

namespace {

size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

DConv make_conv(const std::string& p, int Cin, int Cout, int T0, int T1, int s0, int s1, int p0, int p1, int act) {
    DConv c; c.prefix = p; c.Cin = Cin; c.Cout = Cout; c.CinP = Cin < 16 ? 4 : rup(Cin, 16); c.CoutP = rup(Cout, 16); c.CinR = rup(c.CinP, 16);
    c.T0 = T0; c.T1 = T1; c.s0 = s0; c.s1 = s1; c.p0 = p0; c.p1 = p1; c.act = act;
    c.Kf = rup(T0 * T1 * c.CinP, 16); c.Kt = T0 * T1 * c.CoutP;
    if (s0 * s1 > 1)
        for (int r0 = 0; r0 < s0; ++r0) for (int r1 = 0; r1 < s1; ++r1)
            c.wp_floats += (size_t)c.CinR * phase_ntaps(r0, p0, s0, T0) * phase_ntaps(r1, p1, s1, T1) * c.CoutP;
    return c;
}

int out_dim(int D, int T, int s, int p) { return (D + 2 * p - T) / s + 1; }

// per-sub-discriminator geometry for clips of L samples
struct SubGeom { int D0, D1; std::vector<int> O0, O1; std::vector<int> bandF; int T = 0, catF = 0; };

SubGeom geometry(const DSub& S, int L) {
    SubGeom g;
    if (S.kind == 0) {
        const int p = S.arg;
        g.D0 = (L + (p - L % p)) / p; g.D1 = p;
        int d0 = g.D0;
        for (const DConv& c : S.convs) { d0 = out_dim(d0, c.T0, c.s0, c.p0); g.O0.push_back(d0); g.O1.push_back(p); }
    } else {
        const int w = S.arg, hop = w / 4;
        g.T = (L + hop - 1) / hop;
        const int nb = (int)S.bands.size();
        g.catF = 0;
        for (int b = 0; b < nb; ++b) {
            int f = S.bands[b].second - S.bands[b].first;
            g.bandF.push_back(f);
            for (int j = 0; j < 5; ++j) { const DConv& c = S.convs[b * 5 + j]; f = out_dim(f, c.T1, c.s1, c.p1); g.O0.push_back(g.T); g.O1.push_back(f); }
            g.catF += f;
        }
        g.O0.push_back(g.T); g.O1.push_back(g.catF);
    }
    return g;
}

void fmap_shapes(escx_disc_s* d, int L, std::vector<FmapShape>* out) {
    out->clear();
    for (size_t si = 0; si < d->subs.size(); ++si) {
        const DSub& S = d->subs[si];
        const SubGeom g = geometry(S, L);
        if (S.kind == 0) {
            for (size_t j = 0; j < S.convs.size(); ++j) out->push_back({(int)si, S.convs[j].Cout, S.convs[j].CoutP, g.O0[j], g.O1[j], g.O1[j], 0, -1});
        } else {
            const int nb = (int)S.bands.size();
            int off = 0;
            for (int b = 0; b < nb; ++b)
                for (int j = 0; j < 5; ++j) {
                    const DConv& c = S.convs[b * 5 + j];
                    const bool top = j == 4;
                    out->push_back({(int)si, c.Cout, c.CoutP, g.O0[b * 5 + j], g.O1[b * 5 + j], top ? g.catF : g.O1[b * 5 + j], top ? off : 0, -1});
                    if (top) off += g.O1[b * 5 + j];
                }
            const DConv& c = S.convs.back();
            out->push_back({(int)si, c.Cout, c.CoutP, g.T, g.catF, g.catF, 0, -1});
        }
    }
}

TView view_of(float* base, const FmapShape& f) { return TView{base, f.D0, f.D1, f.P1, f.Cp}; }

// Per-launch-group timing of the convolutions (events + a stream sync after each group: a diagnostic, not a mode to run in).
// ESCX_DISC_TRACE=1 prints every group; escx_disc_profile_enable(1) accumulates them per (kind, layer) for escx_disc_profile_report() - what
// bench.py --mode train_adv quotes its per-kernel roofline from.
struct DProfAgg { int calls = 0; double ms = 0, flops = 0; };
static std::mutex g_dprof_mu;
static bool g_dprof_on = false;
static std::map<std::string, DProfAgg> g_dprof;
static std::vector<std::string> g_dprof_order;
static std::string g_dprof_json;
struct DTrace {
    static bool print() { static const bool v = [] { const char* e = getenv("ESCX_DISC_TRACE"); return e && e[0] == '1'; }(); return v; }
    static bool on() { return print() || g_dprof_on; }
    hipStream_t st; hipEvent_t a = nullptr, b = nullptr; const char* what; const char* name; double flops; int M, N, K;
    DTrace(hipStream_t s, const char* w, const char* nm, int M_, int N_, int K_) : st(s), what(w), name(nm), flops(2.0 * M_ * N_ * K_), M(M_), N(N_), K(K_) {
        if (on()) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~DTrace() {
        if (!a) return;
        (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
        if (print()) fprintf(stderr, "[disc] %-4s %-44s M %8d N %5d K %5d  %8.3f ms  %6.1f TFLOP/s\n", what, name, M, N, K, ms, flops / (ms * 1e9));
        if (g_dprof_on) {
            std::lock_guard<std::mutex> lk(g_dprof_mu);
            const std::string key = std::string("D.") + what + "[" + name + "]";
            if (!g_dprof.count(key)) g_dprof_order.push_back(key);
            DProfAgg& g = g_dprof[key]; g.calls++; g.ms += ms; g.flops += flops;
        }
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
};

// params_version: any number that changes whenever the flat buffer's CONTENTS change (negative = unknown: always re-pack).  The weight-normalised
// operands are rebuilt only then - an adversarial step makes five calls on the same weights, 108 pack launches each otherwise.
thread_local int tls_conv_bf16 = 0;      // precision of the handle whose forward / backward is being enqueued by this thread
thread_local const float* tls_w32 = nullptr; thread_local const __bf16* tls_w16 = nullptr; thread_local size_t tls_wn = 0;      // its packed weights and their bf16 image

int pack_weights(escx_disc_s* d, const float* flat, long long params_version, hipStream_t st) {
    if (params_version >= 0 && params_version == d->packed_version && flat == d->packed_ptr) return 0;
    d->packed_version = params_version; d->packed_ptr = flat;
    for (DSub& S : d->subs)
        for (DConv& c : S.convs)
            hipLaunchKernelGGL(wn_pack_kernel, dim3(c.Cout), dim3(256), 0, st, flat + c.off_v, flat + c.off_g, flat + c.off_b, c.Wf, c.Wt, c.bias, c.scale,
                               c.Cout, c.Cin, c.T0 * c.T1, c.CinP, c.CoutP, c.Kf, c.Kt, c.Wp, c.T0, c.T1, c.s0, c.s1, c.p0, c.p1, c.CinR);
    return launch_ok("disc_pack_weights");
}

// bf16 precision: the packed weights rounded once per parameter version (the convolution kernels would round the same values again for every tile they stage)
int refresh_bf16_weights(escx_disc_s* d, hipStream_t st) {
    tls_conv_bf16 = d->precision; tls_w32 = d->wbuf; tls_wn = d->wfloats; tls_w16 = nullptr;
    static const bool w16_ok = [] { const char* e = ESCX_TUNE_ENV("ESCX_DISC_BF16_WEIGHTS"); return !(e && e[0] == '0'); }();      // 0: weights rounded while staged (A/B)
    if (!d->precision || !w16_ok) return 0;
    if (!d->wbuf16) ESCX_HIP(hipMalloc((void**)&d->wbuf16, 3 * d->wfloats * sizeof(__bf16)));      // precision 2: three planes (the exact three-term split of every weight)
    if (d->w16_ptr != d->packed_ptr || d->w16_version != d->packed_version || d->packed_version < 0 || d->w16_mode != d->precision) {
        if (d->precision == 2) hipLaunchKernelGGL(split3_bf16_kernel, dim3(2048), dim3(256), 0, st, d->wbuf, d->wbuf16, d->wfloats / 4, d->wfloats);
        else hipLaunchKernelGGL(cvt_bf16_kernel, dim3(2048), dim3(256), 0, st, d->wbuf, d->wbuf16, d->wfloats / 4);
        d->w16_ptr = d->packed_ptr; d->w16_version = d->packed_version; d->w16_mode = d->precision;
    }
    tls_w16 = d->wbuf16;
    return launch_ok("disc_bf16_weights");
}

// Multi-stream schedule over the sub-discriminators (they share nothing but the input): the launches of one sub-discriminator fill the dispatch tails and the
// small-grid layers of another.  ESCX_DISC_STREAMS = n (1..4, default 2; 1 = everything on the caller's stream, also while the per-layer profile is on: its
// events bracket single launches).  Sub-discriminator si runs on stream ESCX_DISC_STREAM_MAP[si] (a digit string, default si mod n); stream 0 is the caller's.
int disc_streams() {
    static const int n = [] { const char* e = getenv("ESCX_DISC_STREAMS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    return DTrace::on() ? 1 : n;
}
int disc_stream_of(int si, int n) {
    static const std::string map = [] { const char* e = ESCX_TUNE_ENV("ESCX_DISC_STREAM_MAP"); return std::string(e ? e : ""); }();
    const int q = si < (int)map.size() && map[si] >= '0' && map[si] <= '3' ? map[si] - '0' : si % n;
    return q < n ? q : si % n;
}
int disc_fork(escx_disc_s* d, hipStream_t st, int n) {
    if (!d->ev_fork) ESCX_HIP(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
    ESCX_HIP(hipEventRecord(d->ev_fork, st));
    for (int i = 0; i + 1 < n; ++i) {
        if (!d->aux[i]) {
            ESCX_HIP(hipStreamCreateWithFlags(&d->aux[i], hipStreamNonBlocking));
            ESCX_HIP(hipEventCreateWithFlags(&d->ev_join[i], hipEventDisableTiming));
        }
        ESCX_HIP(hipStreamWaitEvent(d->aux[i], d->ev_fork, 0));
    }
    return 0;
}
int disc_join(escx_disc_s* d, hipStream_t st, int n);
// an error return between fork and join must not leave the aux streams running on scratch and caller tensors that the host is about to
// release.  The guard joins them into the caller's stream (events, no host wait) unless the normal join already ran.
struct DiscJoinGuard {
    escx_disc_s* d; hipStream_t st; int n; bool armed;
    DiscJoinGuard(escx_disc_s* d_, hipStream_t st_, int n_) : d(d_), st(st_), n(n_), armed(n_ > 1) {}
    ~DiscJoinGuard() { if (armed) (void)disc_join(d, st, n); }
    int join() { armed = false; return disc_join(d, st, n); }
};
int disc_join(escx_disc_s* d, hipStream_t st, int n) {
    for (int i = 0; i + 1 < n; ++i) {
        ESCX_HIP(hipEventRecord(d->ev_join[i], d->aux[i]));
        ESCX_HIP(hipStreamWaitEvent(st, d->ev_join[i], 0));
    }
    return 0;
}

// the convolution loaders address a feature map with 32-bit element offsets
int check_map_sizes(const std::vector<FmapShape>& shp, int B) {
    for (const FmapShape& f : shp)
        if ((unsigned long long)B * f.D0 * f.P1 * f.Cp >= (1ull << 32)) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "feature map of %d x %d x %d x %d elements: batch too large for one discriminator pass", B, f.D0, f.P1, f.Cp);
    return 0;
}

int ensure_scratch(escx_disc_s* d, size_t bytes) {
    if (d->scratch_bytes >= bytes) return 0;
    ESCX_HIP(hipDeviceSynchronize());
    if (d->scratch) ESCX_HIP(hipFree(d->scratch));
    d->scratch = nullptr; d->scratch_bytes = 0;
    ESCX_HIP(hipMalloc((void**)&d->scratch, bytes));
    d->scratch_bytes = bytes;
    return 0;
}

inline int conv_cp(const ConvS& l) { return l.x.Cp; }
inline int conv_cp(const ConvTS& l) { return l.y.Cp; }
inline int conv_cp(const ConvTSP& l) { return l.y.Cp; }
inline int conv_cp(const PlainA&) { return 0; }


template <class Ld, class Epi>
void conv_gemm(const Ld& ld_in, const float* W, int M, int Np, int Kp, const Epi& ep, hipStream_t st) {
    Ld ld = ld_in;
    if constexpr (!std::is_same<Ld, PlainA>::value) {            // the 32 -> 32-channel band convolutions: input tile held in LDS (conv32_halo.h), bit-identical to the engine
#ifdef ESCX_EXPERIMENTAL       // fp32 LDS-tile form: bit-identical, 4-10 % faster alone, the step 4 % SLOWER (profiles/r4_disc_ab.txt); tagged builds only
        static const bool halo = [] { const char* e = ESCX_TUNE_ENV("ESCX_CONV32_HALO"); return e && e[0] == '1'; }();
        if (halo && Np == 32 && launch_conv32_halo(make_halo32(ld_in, M, Np, Kp), W, Kp, ep, st)) return;
#endif
        static const bool halo16 = [] { const char* e = ESCX_TUNE_ENV("ESCX_CONV32_HALO_BF16"); return !(e && e[0] == '0'); }();      // bf16 precision: the band convolutions too (A/B: 0)
        if (tls_conv_bf16 == 1 && halo16 && (Np == 32 || Np == 16) && launch_conv32_halo_bf16(make_halo32(ld_in, M, Np, Kp), W, Kp, ep, st)) return;
    }
    if constexpr (!std::is_same<Ld, PlainA>::value) {            // opt-in bf16 MFMA for the wide layers (gemm_bf16.h): same gathers, same epilogues
        if (tls_conv_bf16 && bf16_gemm_ok(M, Np, Kp) && conv_cp(ld) % 32 == 0) {
            const __bf16* W16 = (tls_w16 && W >= tls_w32 && W < tls_w32 + tls_wn) ? tls_w16 + (W - tls_w32) : nullptr;
            ld.fast = 1; launch_gemm_bf16(ld, W, W16, M, Np, Kp, ep, st, tls_conv_bf16 == 2 ? 3 : 1, tls_wn); return;
        }
    }
    const long long tiles128 = (long long)((M + 127) / 128) * ((Np + 95) / 96);
    // K steps of 16: 18 KB of LDS per workgroup instead of 75 KB at the engine's default step of 80 for K = 5 x 1024 - twice the resident
    // workgroups per CU; measured on the step's convolutions (tools/disc_trace.py): forward 73.5 -> 60.2 ms, dX 115.8 -> 99.0 ms
    static const int env_bk = [] { const char* e = ESCX_TUNE_ENV("ESCX_CONV_BK"); return e ? atoi(e) : 16; }();
    static const int env_bkn = [] { const char* e = ESCX_TUNE_ENV("ESCX_CONV_BK_NARROW"); return e ? atoi(e) : 0; }();       // K step of the <= 48-channel outputs (0: the same)
    const int want_bk = (Np <= 48 && env_bkn > 0) ? env_bkn : env_bk;
    const int fbk = (want_bk > 0 && Kp % want_bk == 0) ? want_bk : 0;
    if constexpr (!std::is_same<Ld, PlainA>::value) {            // uniform-tap gathers only when no K step of the engine straddles a tap
        const int bk = fbk ? fbk : pick_bk(Kp), cp = conv_cp(ld);
        ld.fast = (cp > 0 && cp % bk == 0) ? 1 : 0;
    }
    // Narrow outputs (the 32-channel MRD stacks, the first MPD layers): a 128-row tile gives a wave 2 x 2 accumulator tiles - 16 MFMAs per four LDS
    // fragment reads and per K step; 256 rows double the MFMAs per weight fragment and per barrier.  MEASURED SLOWER (round 3, adversarial step at 36
    // clips): MRD band convolutions forward 79 -> 67 TFLOP/s, dX 68 -> 63, step 399.6 -> 402.0 ms (half the workgroups, 4 gather contexts per
    // thread).  Kept as an A/B switch only: ESCX_CONV_BM256=1.
    static const bool bm256 = [] { const char* e = ESCX_TUNE_ENV("ESCX_CONV_BM256"); return e && e[0] == '1'; }();
    if (bm256 && Np <= 48 && (long long)((M + 255) / 256) * ((Np + 47) / 48) >= 1024) { launch_gemm<256>(ld, W, M, Np, Kp, ep, st, 1, fbk); return; }
    if (tiles128 >= 512) launch_gemm<128>(ld, W, M, Np, Kp, ep, st, 1, fbk);
    else launch_gemm<64>(ld, W, M, Np, Kp, ep, st, 1, fbk);
}

ConvS make_convs(const TView& x, const DConv& c, int O0, int O1, int B) {
    ConvGeom g{c.T0, c.T1, c.s0, c.s1, c.p0, c.p1, O0, O1};
    return ConvS{x, g, B * O0 * O1, FastDiv(O0 * O1), FastDiv(O1), FastDiv(x.Cp), FastDiv(c.T1)};
}

void conv_forward(const TView& x, const DConv& c, const TView& out, int B, hipStream_t st) {
    DTrace tr(st, "fwd", c.prefix.c_str(), B * out.D0 * out.D1, c.CoutP, c.Kf);
    ConvSU ld; static_cast<ConvS&>(ld) = make_convs(x, c, out.D0, out.D1, B);
    conv_gemm(ld, c.Wf, B * out.D0 * out.D1, c.CoutP, c.Kf, EpiConvOut{out, c.bias, c.act, FastDiv(out.D0 * out.D1), FastDiv(out.D1)}, st);
}

constexpr size_t DISC_DW_PART = (size_t)48 << 20;          // floats: 192 MB of partial sums (8 slices of the 1024 x 5120 gradient)

// dW partial-sum launch (same scheme as train.hip's dw_launch)
template <class LdA, class LdB>
int disc_dw(const LdA& la, const LdB& lb, int M, int Np, int Kp, float* dW, float* db, float* part, hipStream_t st) {
    const size_t per = (size_t)Np * Kp + Np;
    static const bool wide_ok = [] { const char* e = ESCX_TUNE_ENV("ESCX_DW_WIDE"); return !(e && e[0] == '0'); }();
    const bool big = Np >= 128 && Kp >= 128, narrow = Np == 32 && Kp >= 96;        // narrow: the 32-channel band stacks, from round 4 their 2 -> 32 first layers too (K = 112)
    static const bool dw16n = [] { const char* e = ESCX_TUNE_ENV("ESCX_DISC_BF16_DW_NARROW"); return !(e && e[0] == '0'); }();       // bf16 precision: dW of the band stacks too (A/B: 0)
    static const bool dw_bf16_ok = [] { const char* e = ESCX_TUNE_ENV("ESCX_DISC_BF16_DW"); return !(e && e[0] == '0'); }();      // 0: bf16 precision keeps the fp32 dW kernels (A/B)
    if (tls_conv_bf16 && dw_bf16_ok && Np % 128 == 0 && Kp >= 128 && Kp % 16 == 0) {      // 128 x 128 tiles of dW on the bf16 MFMA (gemm_bf16.h); columns behind Kp read as zero taps
        const int nbn = Np / 128, nbk = (Kp + 127) / 128, blocks = nbn * nbk;
        int slices = std::max(1, std::min((2560 + blocks / 2) / blocks, (M + 255) / 256));
        slices = (int)std::max<size_t>(1, std::min<size_t>(slices, DISC_DW_PART / per));
        int mps = ((M + slices - 1) / slices + 31) / 32 * 32;
        slices = (M + mps - 1) / mps;
        float* bpart = part + (size_t)slices * Np * Kp;
        if (tls_conv_bf16 == 2) hipLaunchKernelGGL((gemm_dw_bf16_kernel<LdA, LdB, 3>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
        else hipLaunchKernelGGL((gemm_dw_bf16_kernel<LdA, LdB>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
        launch_reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
        launch_reduce_partials(bpart, slices, (long long)Np, db, 0, st);
        return 0;
    }
    if (narrow && tls_conv_bf16 == 1 && dw16n && Kp % 16 == 0) {     // 32 x 128 tiles of dW on the bf16 MFMA (gemm_bf16.h)
        const int nbk = (Kp + 127) / 128;
        int slices = std::max(1, std::min((2560 + nbk / 2) / nbk, (M + 255) / 256));
        slices = (int)std::max<size_t>(1, std::min<size_t>(slices, DISC_DW_PART / per));
        int mps = ((M + slices - 1) / slices + 31) / 32 * 32;
        slices = (M + mps - 1) / mps;
        float* bpart = part + (size_t)slices * Np * Kp;
        hipLaunchKernelGGL((gemm_dw_bf16_n32_kernel<LdA, LdB>), dim3(nbk, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
        launch_reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
        launch_reduce_partials(bpart, slices, (long long)Np, db, 0, st);
        return 0;
    }
    if (wide_ok && (big || narrow)) {                         // wide workgroup tiles (gemm_dw3_kernel), ~5 rounds of 2 workgroups per CU
        const int WA = big ? 128 : 32, WB = big ? 128 : 192;
        const int nbn = (Np + WA - 1) / WA, nbk = (Kp + WB - 1) / WB, blocks = nbn * nbk;
        int slices = std::max(1, std::min((2560 + blocks / 2) / blocks, (M + 255) / 256));
        slices = (int)std::max<size_t>(1, std::min<size_t>(slices, DISC_DW_PART / per));
        int mps = ((M + slices - 1) / slices + 31) / 32 * 32;
        slices = (M + mps - 1) / mps;
        float* bpart = part + (size_t)slices * Np * Kp;
        if (big) hipLaunchKernelGGL((gemm_dw3_kernel<LdA, LdB, 4, 4, 2, 2, true>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
        else hipLaunchKernelGGL((gemm_dw3_kernel<LdA, LdB, 2, 3, 1, 4, true>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
        launch_reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
        launch_reduce_partials(bpart, slices, (long long)Np, db, 0, st);
        return 0;
    }
    const int nbn = (Np + 47) / 48, nbk = (Kp + 47) / 48, blocks = nbn * nbk;
    int slices = std::max(1, std::min((2048 + blocks - 1) / blocks, (M + 127) / 128));
    slices = (int)std::min<size_t>(slices, DISC_DW_PART / per);
    if (slices < 1) ESCX_FAIL(ESCX_ERR_STATE, "discriminator dW scratch too small for %d x %d", Np, Kp);
    int mps = ((M + slices - 1) / slices + 127) / 128 * 128;
    slices = (M + mps - 1) / mps;
    float* bpart = part + (size_t)slices * Np * Kp;
    hipLaunchKernelGGL((gemm_dw_kernel<LdA, LdB, true>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
    launch_reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
    launch_reduce_partials(bpart, slices, (long long)Np, db, 0, st);
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
extern "C" int escx_disc_create(const escx_disc_config* cfg, int device, escx_disc* out) {
    if (!cfg || !out) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null argument");
    if (cfg->n_rates != 0) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "MSD (rates) is not implemented: every ESC configuration uses rates = []");
    if (cfg->n_periods < 0 || cfg->n_periods > 8 || cfg->n_ffts < 0 || cfg->n_ffts > 8 || cfg->n_bands < 1 || cfg->n_bands > 8)
        ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad discriminator configuration");
    escx_disc_s* d = new escx_disc_s();
    d->device = device; d->sample_rate = cfg->sample_rate;
    int idx = 0;
    for (int i = 0; i < cfg->n_periods; ++i, ++idx) {
        DSub S; S.kind = 0; S.arg = cfg->periods[i];
        if (S.arg < 1) { delete d; ESCX_FAIL(ESCX_ERR_INVALID_ARG, "period must be positive"); }
        const std::string p = "discriminators." + std::to_string(idx) + ".";
        const int ch[6] = {1, 32, 128, 512, 1024, 1024};
        for (int j = 0; j < 5; ++j) S.convs.push_back(make_conv(p + "convs." + std::to_string(j) + ".0.", ch[j], ch[j + 1], 5, 1, j < 4 ? 3 : 1, 1, 2, 0, 1));
        S.convs.push_back(make_conv(p + "conv_post.", 1024, 1, 3, 1, 1, 1, 1, 0, 0));
        d->subs.push_back(S);
    }
    for (int i = 0; i < cfg->n_ffts; ++i, ++idx) {
        DSub S; S.kind = 1; S.arg = cfg->fft_sizes[i];
        if (S.arg < 16 || S.arg % 16) { delete d; ESCX_FAIL(ESCX_ERR_INVALID_ARG, "fft size must be a positive multiple of 16"); }
        const int nf = S.arg / 2 + 1;
        for (int b = 0; b < cfg->n_bands; ++b) S.bands.push_back({(int)(cfg->bands[b][0] * nf), (int)(cfg->bands[b][1] * nf)});
        const std::string p = "discriminators." + std::to_string(idx) + ".";
        for (int b = 0; b < cfg->n_bands; ++b)
            for (int j = 0; j < 5; ++j) {
                const int k1 = j < 4 ? 9 : 3, s1 = (j >= 1 && j <= 3) ? 2 : 1;
                S.convs.push_back(make_conv(p + "band_convs." + std::to_string(b) + "." + std::to_string(j) + ".0.", j == 0 ? 2 : 32, 32, 3, k1, 1, s1, 1, k1 / 2, 1));
            }
        S.convs.push_back(make_conv(p + "conv_post.", 32, 1, 3, 3, 1, 1, 1, 1, 0));
        S.Fq = rup(nf, 16);
        d->subs.push_back(S);
    }
    // flat parameter layout: the reference's named_parameters() order = per conv (bias, weight_g, weight_v)
    size_t off = 0, wf = 0;
    for (DSub& S : d->subs) {
        for (DConv& c : S.convs) {
            const size_t nv = (size_t)c.Cout * c.Cin * c.T0 * c.T1;
            c.off_b = off; d->keys.push_back(c.prefix + "bias"); d->offs.push_back(off); d->numels.push_back(c.Cout); off += c.Cout;
            c.off_g = off; d->keys.push_back(c.prefix + "weight_g"); d->offs.push_back(off); d->numels.push_back(c.Cout); off += c.Cout;
            c.off_v = off; d->keys.push_back(c.prefix + "weight_v"); d->offs.push_back(off); d->numels.push_back(nv); off += nv;
            wf += pad64((size_t)c.CoutP * c.Kf) + pad64((size_t)c.CinR * c.Kt) + pad64(c.CoutP) + pad64(2 * c.Cout) + pad64(c.wp_floats);
        }
        if (S.kind == 1) wf += 2 * pad64((size_t)2 * S.Fq * S.arg);
    }
    d->total = off;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete d; ESCX_FAIL(ESCX_ERR_HIP, "hipSetDevice failed"); }
    if (hipMalloc((void**)&d->wbuf, wf * sizeof(float)) != hipSuccess) { delete d; ESCX_FAIL(ESCX_ERR_HIP, "hipMalloc of the discriminator weights failed"); }
    (void)hipMemset(d->wbuf, 0, wf * sizeof(float));
    d->wfloats = wf;
    size_t cur = 0;
    auto take = [&](size_t n) { float* p = d->wbuf + cur; cur += pad64(n); return p; };
    for (DSub& S : d->subs) {
        for (DConv& c : S.convs) { c.Wf = take((size_t)c.CoutP * c.Kf); c.Wt = take((size_t)c.CinR * c.Kt); c.bias = take(c.CoutP); c.scale = take(2 * c.Cout);
                                   if (c.wp_floats) c.Wp = take(c.wp_floats); }
        if (S.kind == 1) {      // windowed DFT of AudioSignal.stft: periodic hann of the full window length
            const int w = S.arg, nf = w / 2 + 1, Fq = S.Fq;
            S.D = take((size_t)2 * Fq * w); S.DT = take((size_t)2 * Fq * w);
            std::vector<float> hD((size_t)2 * Fq * w, 0.f), hT((size_t)2 * Fq * w, 0.f);
            for (int f = 0; f < nf; ++f) for (int k = 0; k < w; ++k) {
                const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * k / w);
                const double ang = 2.0 * M_PI * (double)((long long)f * k % w) / w;
                const float re = (float)(win * std::cos(ang)), im = (float)(-win * std::sin(ang));
                hD[(size_t)f * w + k] = re; hD[(size_t)(Fq + f) * w + k] = im;
                hT[(size_t)k * 2 * Fq + f] = re; hT[(size_t)k * 2 * Fq + Fq + f] = im;
            }
            (void)hipMemcpy(S.D, hD.data(), hD.size() * sizeof(float), hipMemcpyHostToDevice);
            (void)hipMemcpy(S.DT, hT.data(), hT.size() * sizeof(float), hipMemcpyHostToDevice);
        }
    }
    *out = d;
    return ESCX_OK;
}

extern "C" void escx_disc_destroy(escx_disc d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->wbuf) (void)hipFree(d->wbuf);
    if (d->wbuf16) (void)hipFree(d->wbuf16);
    if (d->scratch) (void)hipFree(d->scratch);
    for (int i = 0; i < 3; ++i) {
        if (d->aux[i]) { (void)hipStreamSynchronize(d->aux[i]); (void)hipStreamDestroy(d->aux[i]); }
        if (d->ev_join[i]) (void)hipEventDestroy(d->ev_join[i]);
    }
    if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
    delete d;
}

extern "C" int escx_disc_set_precision(escx_disc d, int mode) {
    if (!d) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    if (mode < 0 || mode > 2)
        ESCX_FAIL(ESCX_ERR_INVALID_ARG, "discriminator precision %d: 0 (fp32 MFMA), 1 (bf16 MFMA for the convolutions) or 2 (split operands: fp32-grade on the bf16 MFMA, wide convolutions)", mode);
    d->precision = mode;
    return ESCX_OK;
}
extern "C" int escx_disc_get_precision(escx_disc d) { return d ? d->precision : -1; }

extern "C" int escx_disc_param_count(escx_disc d) { return d ? (int)d->keys.size() : 0; }
extern "C" const char* escx_disc_param_key(escx_disc d, int i) { return (d && i >= 0 && i < (int)d->keys.size()) ? d->keys[i].c_str() : nullptr; }
extern "C" int64_t escx_disc_param_offset(escx_disc d, int i) { return (d && i >= 0 && i < (int)d->offs.size()) ? (int64_t)d->offs[i] : -1; }
extern "C" int64_t escx_disc_param_numel(escx_disc d, int i) { return (d && i >= 0 && i < (int)d->numels.size()) ? (int64_t)d->numels[i] : -1; }
extern "C" int64_t escx_disc_param_total(escx_disc d) { return d ? (int64_t)d->total : 0; }

// Feature maps of one forward for clips of n_samples: count, and for map i its sub-discriminator, channels, padded channels, extent (D0, D1), the
// row pitch P1 and column offset of the buffer it lives in (the five band tops of an MRD are slices of one [T][sum F][32] buffer).
extern "C" int escx_disc_num_fmaps(escx_disc d, int n_samples) {
    if (!d) return 0;
    std::vector<FmapShape> f; fmap_shapes(d, n_samples, &f);
    return (int)f.size();
}
extern "C" int escx_disc_fmap_shape(escx_disc d, int n_samples, int i, int* sub, int* C, int* Cp, int* D0, int* D1, int* P1, int* off1) {
    if (!d) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    std::vector<FmapShape> f; fmap_shapes(d, n_samples, &f);
    if (i < 0 || i >= (int)f.size()) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "feature map index out of range");
    if (sub) *sub = f[i].sub; if (C) *C = f[i].C; if (Cp) *Cp = f[i].Cp; if (D0) *D0 = f[i].D0; if (D1) *D1 = f[i].D1; if (P1) *P1 = f[i].P1; if (off1) *off1 = f[i].off1;
    return ESCX_OK;
}

namespace {
// scratch layout shared by forward and backward: pre-processed wave, per-sub input maps / spectrogram rows
struct Front { float* y; float* stats; std::vector<float*> in; std::vector<float*> spec; size_t floats; };

size_t front_floats(escx_disc_s* d, int B, int L) {
    size_t n = pad64((size_t)B * L) + pad64((size_t)4 * B);
    for (const DSub& S : d->subs) {
        const SubGeom g = geometry(S, L);
        if (S.kind == 0) n += pad64((size_t)B * g.D0 * g.D1 * 4);
        else { n += pad64((size_t)B * g.T * 2 * S.Fq); for (int f : g.bandF) n += pad64((size_t)B * g.T * f * 4); }
    }
    return n;
}

// pre-process + build every sub-discriminator's input map (MPD: padded/reshaped wave; MRD: band slices of the matched-stride STFT)
void build_front(escx_disc_s* d, const float* wave, int B, int L, float* base, std::vector<std::vector<float*>>* ins, std::vector<float*>* specs, float** y_out,
                 float** stats_out, hipStream_t st) {
    float* cur = base;
    auto take = [&](size_t n) { float* p = cur; cur += pad64(n); return p; };
    float* y = take((size_t)B * L); float* stats = take((size_t)4 * B);
    hipLaunchKernelGGL(disc_preprocess_kernel, dim3(B), dim3(1024), 0, st, wave, y, stats, L);
    ins->clear(); specs->clear();
    for (const DSub& S : d->subs) {
        const SubGeom g = geometry(S, L);
        std::vector<float*> v;
        if (S.kind == 0) {
            float* in = take((size_t)B * g.D0 * g.D1 * 4);
            hipLaunchKernelGGL(mpd_input_kernel, dim3(blk((long long)B * g.D0 * g.D1)), dim3(256), 0, st, y, in, B, L, g.D0, S.arg);
            v.push_back(in); specs->push_back(nullptr);
        } else {
            const int w = S.arg, hop = w / 4;
            float* spec = take((size_t)B * g.T * 2 * S.Fq);
            gemm_frames(y, B, L, g.T, hop, -(w - hop) / 2, S.D, 2 * S.Fq, w, spec, st);
            for (size_t b = 0; b < S.bands.size(); ++b) {
                float* in = take((size_t)B * g.T * g.bandF[b] * 4);
                hipLaunchKernelGGL(mrd_band_kernel, dim3(blk((long long)B * g.T * g.bandF[b])), dim3(256), 0, st, spec, in, (long long)B * g.T, S.Fq, S.bands[b].first, g.bandF[b]);
                v.push_back(in);
            }
            specs->push_back(spec);
        }
        ins->push_back(v);
    }
    *y_out = y; *stats_out = stats;
}
}  // namespace

// Discriminator.forward (discriminator.py:218-221).  fmaps: host array of escx_disc_num_fmaps() device pointers, each the BASE address of map i
// (i.e. already offset to its column slice when it lives in a concatenated buffer), laid out [B][D0][P1][Cp].
extern "C" int escx_disc_forward(escx_disc d, const float* flat_params, int64_t params_version, const float* wave, int B, int L, float* const* fmaps, void* stream) {
    if (!d || !flat_params || !wave || !fmaps || B < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    ESCX_HIP(hipSetDevice(d->device));
    hipStream_t st = (hipStream_t)stream;
    tls_conv_bf16 = d->precision;
    int maxp = 1; for (const DSub& S : d->subs) if (S.kind == 0) maxp = std::max(maxp, S.arg);
    if (L <= maxp + 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "clip too short for the reflect padding of the period discriminators");
    std::vector<FmapShape> shp; fmap_shapes(d, L, &shp);
    int rc = check_map_sizes(shp, B); if (rc) return rc;
    if ((rc = ensure_scratch(d, front_floats(d, B, L) * sizeof(float)))) return rc;
    if ((rc = pack_weights(d, flat_params, (long long)params_version, st))) return rc;
    if ((rc = refresh_bf16_weights(d, st))) return rc;
    std::vector<std::vector<float*>> ins; std::vector<float*> specs; float *y, *stats;
    build_front(d, wave, B, L, d->scratch, &ins, &specs, &y, &stats, st);
    const int nq = disc_streams();
    if (nq > 1 && (rc = disc_fork(d, st, nq))) return rc;
    const hipStream_t st0 = st;
    DiscJoinGuard guard(d, st0, nq);
    int fi = 0;
    for (size_t si = 0; si < d->subs.size(); ++si) {
        const DSub& S = d->subs[si];
        const SubGeom g = geometry(S, L);
        { const int q = disc_stream_of((int)si, nq); st = q ? d->aux[q - 1] : st0; }
        if (S.kind == 0) {
            TView x{ins[si][0], g.D0, g.D1, g.D1, 4};
            for (size_t j = 0; j < S.convs.size(); ++j, ++fi) {
                TView o = view_of(fmaps[fi], shp[fi]);
                conv_forward(x, S.convs[j], o, B, st);
                x = o;
            }
        } else {
            const int nb = (int)S.bands.size();
            float* cat_base = nullptr;
            for (int b = 0; b < nb; ++b) {
                TView x{ins[si][b], g.T, g.bandF[b], g.bandF[b], 4};
                for (int j = 0; j < 5; ++j, ++fi) {
                    TView o = view_of(fmaps[fi], shp[fi]);
                    conv_forward(x, S.convs[b * 5 + j], o, B, st);
                    x = o;
                    if (j == 4 && b == 0) cat_base = fmaps[fi];
                }
            }
            TView cat{cat_base, g.T, g.catF, g.catF, 32};
            TView o = view_of(fmaps[fi], shp[fi]);
            conv_forward(cat, S.convs.back(), o, B, st);
            ++fi;
        }
    }
    st = st0;
    if (nq > 1 && (rc = guard.join())) return rc;
    return launch_ok("disc_forward");
}

// Backward of the last forward on the same inputs.  d_fmaps[i]: gradient of map i with the SAME layout as fmaps[i] (NULL = zero).
// grad_flat (optional): d loss / d parameters, overwritten.  d_wave (optional): d loss / d waveform (B, L), overwritten.
extern "C" int escx_disc_backward(escx_disc d, const float* flat_params, int64_t params_version, const float* wave, int B, int L, float* const* fmaps,
                                  const float* const* d_fmaps, float* grad_flat, float* d_wave, void* stream) {
    if (!d || !flat_params || !wave || !fmaps || !d_fmaps || B < 1 || (!grad_flat && !d_wave)) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    ESCX_HIP(hipSetDevice(d->device));
    hipStream_t st = (hipStream_t)stream;
    tls_conv_bf16 = d->precision;
    std::vector<FmapShape> shp; fmap_shapes(d, L, &shp);
    // scratch: front + gradient buffers of every feature-map BUFFER (concatenated buffers once) + input-map gradients + dW staging
    const size_t front = front_floats(d, B, L);
    size_t gfl = 0, max_w = 0, in_g = 0;
    {
        int fi = 0;
        for (const DSub& S : d->subs) {
            const SubGeom g = geometry(S, L);
            for (size_t j = 0; j < S.convs.size(); ++j, ++fi) {
                const FmapShape& f = shp[fi];
                const bool slice = f.P1 != f.D1;
                if (!slice) gfl += pad64((size_t)B * f.D0 * f.D1 * f.Cp);
                else if (f.off1 == 0) gfl += pad64((size_t)B * f.D0 * f.P1 * f.Cp);
                max_w = std::max(max_w, (size_t)S.convs[j].CoutP * S.convs[j].Kf + S.convs[j].CoutP);
            }
            if (S.kind == 0) in_g = std::max(in_g, (size_t)B * g.D0 * g.D1 * 4);
            else { for (int f : g.bandF) in_g = std::max(in_g, (size_t)B * g.T * f * 4); in_g = std::max(in_g, (size_t)B * g.T * std::max(2 * S.Fq, S.arg)); }
        }
    }
    const int nq = disc_streams();                              // per-stream copies of the per-layer scratch
    const size_t total = front + gfl + nq * (3 * pad64(in_g) + pad64(max_w) + pad64(DISC_DW_PART) + pad64((size_t)B * L)) + 4096;
    int rc = check_map_sizes(shp, B); if (rc) return rc;
    if ((rc = ensure_scratch(d, total * sizeof(float)))) return rc;
    if ((rc = pack_weights(d, flat_params, (long long)params_version, st))) return rc;
    if ((rc = refresh_bf16_weights(d, st))) return rc;
    std::vector<std::vector<float*>> ins; std::vector<float*> specs; float *y, *stats;
    build_front(d, wave, B, L, d->scratch, &ins, &specs, &y, &stats, st);
    float* cur = d->scratch + front;
    auto take = [&](size_t n) { float* p = cur; cur += pad64(n); return p; };
    // fused[i]: the dX launches of map i's consumer cover every position of the map exactly once, so their epilogue writes the final pre-activation
    // gradient (loss gradient + dX, times the LeakyReLU mask) in one pass; otherwise copy the loss gradient first, accumulate, mask in place.
    static const bool fuse_on = !(getenv("ESCX_DISC_FUSED_GRAD") && atoi(getenv("ESCX_DISC_FUSED_GRAD")) == 0);
    auto dx_covers = [&](const DConv& c) {
        if (!fuse_on) return false;
        if (!c.Wp) return true;
        for (int r0 = 0; r0 < c.s0; ++r0) for (int r1 = 0; r1 < c.s1; ++r1)
            if (phase_ntaps(r0, c.p0, c.s0, c.T0) * phase_ntaps(r1, c.p1, c.s1, c.T1) == 0) return false;
        return true;
    };
    std::vector<char> fused(shp.size(), 0);
    {
        int f0 = 0;
        for (const DSub& S : d->subs) {
            const int nconv = (int)S.convs.size();
            if (S.kind == 0) { for (int j = 0; j + 1 < nconv; ++j) fused[f0 + j] = dx_covers(S.convs[j + 1]); }
            else {
                const int nb = (int)S.bands.size();
                for (int b = 0; b < nb; ++b) for (int j = 0; j < 4; ++j) fused[f0 + b * 5 + j] = dx_covers(S.convs[b * 5 + j + 1]);
                bool cat_ok = dx_covers(S.convs.back());        // the band tops are one concatenated map for conv_post: their loss gradients must be too
                const float* base = d_fmaps[f0 + 4];
                for (int b = 0; b < nb && cat_ok; ++b) {
                    const int fi = f0 + b * 5 + 4;
                    cat_ok = base ? d_fmaps[fi] == base + (size_t)shp[fi].off1 * shp[fi].Cp : d_fmaps[fi] == nullptr;
                }
                for (int b = 0; b < nb; ++b) fused[f0 + b * 5 + 4] = cat_ok;
            }
            f0 += nconv;
        }
    }
    // gradient views, one per feature map (same layout as the map)
    std::vector<TView> gv(shp.size());
    {
        float* cat = nullptr;
        for (size_t i = 0; i < shp.size(); ++i) {
            const FmapShape& f = shp[i];
            if (f.P1 == f.D1) gv[i] = TView{take((size_t)B * f.D0 * f.D1 * f.Cp), f.D0, f.D1, f.D1, f.Cp};
            else { if (f.off1 == 0) cat = take((size_t)B * f.D0 * f.P1 * f.Cp); gv[i] = TView{cat + (size_t)f.off1 * f.Cp, f.D0, f.D1, f.P1, f.Cp}; }
            if (fused[i]) continue;
            TView src{const_cast<float*>(d_fmaps[i]), f.D0, f.D1, f.P1, f.Cp};
            hipLaunchKernelGGL(view_copy_kernel, dim3(blk((long long)B * f.D0 * f.D1 * f.Cp / 4)), dim3(256), 0, st, gv[i], src, (long long)B * f.D0 * f.D1 * f.Cp / 4);
        }
    }
    float *gin_q[4], *gspec_q[4], *gfr_q[4], *dWs_q[4], *part_q[4], *dy_q[4];
    for (int q = 0; q < nq; ++q) {
        gin_q[q] = take(in_g); gspec_q[q] = take(in_g); gfr_q[q] = take(in_g);
        dWs_q[q] = take(max_w); part_q[q] = take(DISC_DW_PART); dy_q[q] = take((size_t)B * L);
    }
    if (grad_flat) ESCX_HIP(hipMemsetAsync(grad_flat, 0, d->total * sizeof(float), st));
    if (d_wave) for (int q = 0; q < nq; ++q) ESCX_HIP(hipMemsetAsync(dy_q[q], 0, (size_t)B * L * sizeof(float), st));
    if (nq > 1 && (rc = disc_fork(d, st, nq))) return rc;
    const hipStream_t st0 = st;
    DiscJoinGuard guard(d, st0, nq);
    float *gin = gin_q[0], *gspec = gspec_q[0], *gfr = gfr_q[0], *dWs = dWs_q[0], *part = part_q[0], *dy = dy_q[0];      // re-pointed per sub-discriminator

    // g: gradient of the layer's output map (g_pre: already the pre-activation gradient).  gx: gradient view of its input map; fin = 1 writes it
    // finally as (init + dX) * LeakyReLU'(xact) (see GradOut).  gx_plain: dense gradient of a network input (no loss gradient, no activation).
    auto layer_bwd = [&](const DConv& c, const TView& x, const TView& yv, const TView& g, bool g_pre, const TView* gx, float* gx_plain, bool fin, const float* init,
                         const float* xact) -> int {
        const int M = B * yv.D0 * yv.D1;
        if (c.act && !g_pre) hipLaunchKernelGGL(leaky_bwd_kernel, dim3(blk((long long)M * yv.Cp / 4)), dim3(256), 0, st, g, yv, (long long)M * yv.Cp / 4, B);
        if (grad_flat) {
            ViewRowsA la{g, M, FastDiv(yv.D0 * yv.D1), FastDiv(yv.D1)};
            ConvS lb = make_convs(x, c, yv.D0, yv.D1, B);
            int r;
            // the bias gradient is reduced straight into the flat gradient buffer where the channel count needs no padding (every layer but the 1-channel heads:
            // 100 copy-engine packets per backward less)
            static const bool db_direct_ok = [] { const char* e = ESCX_TUNE_ENV("ESCX_DISC_DB_DIRECT"); return !(e && e[0] == '0'); }();
            const bool db_direct = db_direct_ok && c.Cout == c.CoutP;
            float* db = db_direct ? grad_flat + c.off_b : dWs + (size_t)c.CoutP * c.Kf;
            { DTrace tr(st, "dW", c.prefix.c_str(), M, c.CoutP, c.Kf); r = disc_dw(la, lb, M, c.CoutP, c.Kf, dWs, db, part, st); }
            if (r) return r;
            hipLaunchKernelGGL(wn_bwd_kernel, dim3(c.Cout), dim3(256), 0, st, dWs, flat_params + c.off_v, c.scale, grad_flat + c.off_v, grad_flat + c.off_g, c.Cin,
                               c.T0 * c.T1, c.CinP, c.Kf);
            if (!db_direct) ESCX_HIP(hipMemcpyAsync(grad_flat + c.off_b, db, (size_t)c.Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        if (gx || gx_plain) {
            TView gxv = gx ? *gx : TView{gx_plain, x.D0, x.D1, x.D1, x.Cp};
            DTrace tr(st, "dX", c.prefix.c_str(), B * x.D0 * x.D1, c.CinR, c.Wp ? c.Kt / (c.s0 * c.s1) : c.Kt);
            if (gx_plain) { fin = dx_covers(c); init = nullptr; xact = nullptr; }
            if (gx_plain && !fin) ESCX_HIP(hipMemsetAsync(gx_plain, 0, (size_t)B * x.D0 * x.D1 * x.Cp * sizeof(float), st));
            const GradOut go{gxv, init, xact, fin ? 1 : 0};
            if (c.Wp) {                         // strided: one launch per residue class of input positions over its own taps
                size_t off = 0;
                for (int r0 = 0; r0 < c.s0; ++r0) for (int r1 = 0; r1 < c.s1; ++r1) {
                    const int n0 = phase_ntaps(r0, c.p0, c.s0, c.T0), n1 = phase_ntaps(r1, c.p1, c.s1, c.T1);
                    const int kp = n0 * n1 * c.CoutP;
                    const int Q0 = x.D0 > r0 ? (x.D0 - r0 + c.s0 - 1) / c.s0 : 0, Q1 = x.D1 > r1 ? (x.D1 - r1 + c.s1 - 1) / c.s1 : 0;
                    if (kp > 0 && Q0 > 0 && Q1 > 0) {
                        PhaseGeom pg{r0, r1, n0, n1, (r0 + c.p0 - phase_tmin(r0, c.p0, c.s0)) / c.s0, (r1 + c.p1 - phase_tmin(r1, c.p1, c.s1)) / c.s1, Q0, Q1, c.s0, c.s1};
                        const int Mi = B * Q0 * Q1;
                        ConvTSP lt{g, pg, Mi, FastDiv(Q0 * Q1), FastDiv(Q1), FastDiv(g.Cp), FastDiv(n1)};
                        conv_gemm(lt, c.Wp + off, Mi, c.CinR, kp, EpiAccumPhase{go, pg, FastDiv(Q0 * Q1), FastDiv(Q1)}, st);
                    }
                    off += (size_t)c.CinR * kp;
                }
            } else {
                ConvGeom cg{c.T0, c.T1, c.s0, c.s1, c.p0, c.p1, yv.D0, yv.D1};
                const int Mi = B * x.D0 * x.D1;
                ConvTS lt{g, cg, x.D0, x.D1, Mi, FastDiv(x.D0 * x.D1), FastDiv(x.D1), FastDiv(g.Cp), FastDiv(c.T1), FastDiv(c.s0), FastDiv(c.s1)};
                conv_gemm(lt, c.Wt, Mi, c.CinR, c.Kt, EpiAccumView{go, FastDiv(x.D0 * x.D1), FastDiv(x.D1)}, st);
            }
        }
        return 0;
    };

    int fi_end = (int)shp.size();
    for (int si = (int)d->subs.size() - 1; si >= 0; --si) {
        const DSub& S = d->subs[si];
        const SubGeom g = geometry(S, L);
        const int nconv = (int)S.convs.size();
        const int f0 = fi_end - nconv;
        const int q = disc_stream_of(si, nq);
        st = q ? d->aux[q - 1] : st0;
        gin = gin_q[q]; gspec = gspec_q[q]; gfr = gfr_q[q]; dWs = dWs_q[q]; part = part_q[q]; dy = dy_q[q];
        if (S.kind == 0) {
            for (int j = nconv - 1; j >= 0; --j) {
                const int fi = f0 + j;
                TView yv = view_of(fmaps[fi], shp[fi]);
                TView x = j > 0 ? view_of(fmaps[fi - 1], shp[fi - 1]) : TView{ins[si][0], g.D0, g.D1, g.D1, 4};
                if (j > 0) {
                    if ((rc = layer_bwd(S.convs[j], x, yv, gv[fi], fused[fi], &gv[fi - 1], nullptr, fused[fi - 1], d_fmaps[fi - 1], S.convs[j - 1].act ? fmaps[fi - 1] : nullptr)))
                        return rc;
                } else { if ((rc = layer_bwd(S.convs[j], x, yv, gv[fi], fused[fi], nullptr, d_wave ? gin : nullptr, false, nullptr, nullptr))) return rc; }
            }
            if (d_wave) hipLaunchKernelGGL(mpd_input_bwd_kernel, dim3(blk((long long)B * L)), dim3(256), 0, st, gin, dy, B, L, g.D0, S.arg);
        } else {
            const int nb = (int)S.bands.size();
            const int fpost = f0 + nconv - 1, ftop0 = f0 + 4;
            TView cat_y{fmaps[ftop0], g.T, g.catF, g.catF, 32};
            TView cat_g{gv[ftop0].p, g.T, g.catF, g.catF, 32};
            if ((rc = layer_bwd(S.convs.back(), cat_y, view_of(fmaps[fpost], shp[fpost]), gv[fpost], fused[fpost], &cat_g, nullptr, fused[ftop0], d_fmaps[ftop0],
                                S.convs[4].act ? fmaps[ftop0] : nullptr)))
                return rc;
            if (d_wave) ESCX_HIP(hipMemsetAsync(gspec, 0, (size_t)B * g.T * 2 * S.Fq * sizeof(float), st));
            for (int b = nb - 1; b >= 0; --b) {
                for (int j = 4; j >= 0; --j) {
                    const int fi = f0 + b * 5 + j;
                    TView yv = view_of(fmaps[fi], shp[fi]);
                    TView x = j > 0 ? view_of(fmaps[fi - 1], shp[fi - 1]) : TView{ins[si][b], g.T, g.bandF[b], g.bandF[b], 4};
                    if (j > 0) {
                        if ((rc = layer_bwd(S.convs[b * 5 + j], x, yv, gv[fi], fused[fi], &gv[fi - 1], nullptr, fused[fi - 1], d_fmaps[fi - 1],
                                            S.convs[b * 5 + j - 1].act ? fmaps[fi - 1] : nullptr)))
                            return rc;
                    } else { if ((rc = layer_bwd(S.convs[b * 5 + j], x, yv, gv[fi], fused[fi], nullptr, d_wave ? gin : nullptr, false, nullptr, nullptr))) return rc; }
                }
                if (d_wave) hipLaunchKernelGGL(mrd_band_bwd_kernel, dim3(blk((long long)B * g.T * g.bandF[b])), dim3(256), 0, st, gin, gspec, (long long)B * g.T, S.Fq,
                                               S.bands[b].first, g.bandF[b]);
            }
            if (d_wave) {
                const int w = S.arg, hop = w / 4;
                PlainA ld{gspec, 2 * S.Fq, B * g.T};
                conv_gemm(ld, S.DT, B * g.T, w, 2 * S.Fq, EpiStore{gfr, w, nullptr}, st);
                hipLaunchKernelGGL(frames_bwd_kernel, dim3(blk((long long)B * L)), dim3(256), 0, st, gfr, dy, B, L, g.T, hop, w, w, 1, (w - hop) / 2);
            }
        }
        fi_end = f0;
    }
    st = st0; dy = dy_q[0];
    if (nq > 1) {
        if ((rc = guard.join())) return rc;
        // the waveform gradients of the streams' sub-discriminators, added in a fixed order
        if (d_wave) for (int q = 1; q < nq; ++q) hipLaunchKernelGGL(add_into_kernel, dim3(blk((long long)B * L)), dim3(256), 0, st, dy, dy_q[q], (long long)B * L);
    }
    if (d_wave) hipLaunchKernelGGL(disc_preprocess_bwd_kernel, dim3(B), dim3(1024), 0, st, dy, y, stats, d_wave, L);
    return launch_ok("disc_backward");
}

// One GAN loss term over a feature map (gan_loss.py:30-51): loss_dev[b] (+)= mean over the map's real elements; grad (optional, same layout as x).
//   mode 0: (target - x)^2 ; mode 1: |x - ref|
extern "C" int escx_gan_term(const float* x, const float* ref, float* grad, int B, int C, int Cp, int D0, int D1, int P1, int mode, float target, float* loss_dev,
                             int accumulate, void* stream) {
    if (!x || !loss_dev || B < 1 || (mode == 1 && !ref) || mode < 0 || mode > 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    TView xv{const_cast<float*>(x), D0, D1, P1, Cp}, rv{const_cast<float*>(ref), D0, D1, P1, Cp}, gvw{grad, D0, D1, P1, Cp};
    const long long per = (long long)D0 * D1 * Cp;
    const int bpc = (int)std::min<long long>(64, (per / 4 + 255) / 256);
    float* part = stream_scratch(st, 0, (size_t)B * bpc);
    if (!part) ESCX_FAIL(ESCX_ERR_HIP, "scratch allocation failed");
    hipLaunchKernelGGL(gan_term_kernel, dim3(bpc, B), dim3(256), 0, st, xv, rv, gvw, part, mode, target, C, bpc, 1.0f / ((float)C * D0 * D1), (const float*)nullptr);
    hipLaunchKernelGGL(row_sum_kernel, dim3(blk(B, 64)), dim3(64), 0, st, part, bpc, loss_dev, B, accumulate, 1.0f);
    return launch_ok("gan_term");
}

// Backward of escx_gan_term with the upstream gradient folded in: grad (layout of x) = g[b] * d term_b / d x, nothing else read or written.
// The forward then keeps no gradient buffer: the term is re-evaluated here (x and ref are still there), one pass instead of store + scale.
extern "C" int escx_gan_term_grad(const float* x, const float* ref, const float* g, float* grad, int B, int C, int Cp, int D0, int D1, int P1, int mode, float target,
                                  void* stream) {
    if (!x || !g || !grad || B < 1 || (mode == 1 && !ref) || mode < 0 || mode > 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    TView xv{const_cast<float*>(x), D0, D1, P1, Cp}, rv{const_cast<float*>(ref), D0, D1, P1, Cp}, gvw{grad, D0, D1, P1, Cp};
    const long long per = (long long)D0 * D1 * Cp;
    const int bpc = (int)std::min<long long>(64, (per / 4 + 255) / 256);
    hipLaunchKernelGGL(gan_term_kernel, dim3(bpc, B), dim3(256), 0, st, xv, rv, gvw, (float*)nullptr, mode, target, C, bpc, 1.0f / ((float)C * D0 * D1), g);
    return launch_ok("gan_term_grad");
}

extern "C" int escx_disc_profile_enable(int enable) {
    std::lock_guard<std::mutex> lk(g_dprof_mu);
    if (enable) { g_dprof.clear(); g_dprof_order.clear(); }
    g_dprof_on = enable != 0;
    return ESCX_OK;
}

extern "C" const char* escx_disc_profile_report() {
    std::lock_guard<std::mutex> lk(g_dprof_mu);
    std::string js = "[";
    char buf[512];
    for (size_t i = 0; i < g_dprof_order.size(); ++i) {
        const DProfAgg& g = g_dprof[g_dprof_order[i]];
        snprintf(buf, sizeof(buf), "%s{\"name\":\"%s\",\"calls\":%d,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":0}", i ? "," : "", g_dprof_order[i].c_str(), g.calls, g.ms, g.flops);
        js += buf;
    }
    g_dprof_json = js + "]";
    return g_dprof_json.c_str();
}
