// libescx host side, part 3 of 3: the launch SEQUENCES of the inference path (one TransformerLayer, encoder, cross-scale quantiser, decoder), the derived weight images of the
// precision modes, and the C-ABI entry points that run them (include/escx.h).  Packing: escx_params.cpp; profiler: escx_profile.cpp.
// libescx C ABI implementation: handle, parameter packing, workspace, and the launch sequences of
// ESC.encode / ESC.decode / ESC.forward(eval).  Reference citations are relative to /root/reference/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "escx_internal.h"
#include "launchers.h"

using namespace escx;

// Whole-path calls: run `part(first_clip, n_clips, stream)` once, or as two halves on two streams joined by events.
// Clips are independent end to end, so the halves never exchange data; overlapping them lets one half's kernels fill
// the CUs the other half's tail workgroups leave idle (a 36-clip layer launches only ~1.3-2.6 workgroups per CU).
template <class F>
static int run_halves(escx_handle_s* h, int B, int T, hipStream_t st, F part) {
    const int k = std::min(n_parts(h, B), h->n_sets);
    // a part walks its clips in passes of at most `pass` clips (escx_params.cpp pass_clips: the workspace sets are sized for one pass)
    const int pass = std::max(1, std::min(pass_clips(h, B, T), h->sets[0].shp.B));
    auto run_part = [&](int b0, int nb, hipStream_t si) -> int {
        for (int off = 0; off < nb; off += pass) { const int r = part(b0 + off, std::min(pass, nb - off), si); if (r) return r; }
        return 0;
    };
    if (k <= 1) { use_set(h, 0); return run_part(0, B, st); }
    const int per = (B + k - 1) / k;
    // Event timing stays valid under concurrency (each kernel is bracketed on its own stream); ESCX_PROF_SERIAL=1 puts the
    // parts back to back on the caller's stream when isolated per-kernel durations are wanted.
    const bool concurrent = !(h->prof && (h->prof_serial || h->prof_isolated));
    if (concurrent) ESCX_HIP(hipEventRecord(h->ev_fork, st));
    // Join guard: once a part has been enqueued on a side stream, NO error path may return before the caller's stream waits
    // for it - the caller is free to release or reuse the buffers the moment this function returns an error.  Failures between fork and join are
    // therefore collected, every started side stream is joined (event, or a blocking stream synchronise when the event cannot be recorded), and
    // only then is the first failure reported.
    int rc = 0;
    bool started[escx_handle_s::MAX_PARTS] = {};
    auto hip_fail = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == 0) { set_error("run_halves: %s: %s", what, hipGetErrorString(e)); rc = ESCX_ERR_HIP; }
        return e != hipSuccess;
    };
    // fork every side stream, then enqueue the passes ROUND-ROBIN over the parts (pass j of part 0, pass j of part 1, ...): the host feeds both streams from the
    // start instead of queueing a whole part (hundreds of launches for a 288-clip batch) before the other stream sees its first kernel
    for (int i = 1; i < k && concurrent && !rc; ++i) {
        if (i * per >= B) break;
        if (hip_fail(hipStreamWaitEvent(h->sx[i], h->ev_fork, 0), "fork wait")) break;
        started[i] = true;
    }
    for (int off = 0; off < per && !rc; off += pass)
        for (int i = 0; i < k && !rc; ++i) {
            const int b0 = i * per, nb = std::min(per, B - b0);
            if (off >= nb) continue;
            const bool side = concurrent && i > 0;
            if (side && !started[i]) continue;
            use_set(h, i);
            const int prc = part(b0 + off, std::min(pass, nb - off), side ? h->sx[i] : st);
            if (prc && !rc) rc = prc;
        }
    for (int i = 1; i < k; ++i) {
        if (!started[i]) continue;
        if (hipEventRecord(h->ev_join[i], h->sx[i]) != hipSuccess || hipStreamWaitEvent(st, h->ev_join[i], 0) != hipSuccess) {
            const hipError_t e = hipStreamSynchronize(h->sx[i]);          // cannot order the streams with an event: wait on the host instead
            hip_fail(e == hipSuccess ? hipErrorUnknown : e, "join");
        }
    }
    use_set(h, 0);
    return rc;
}

static int frames_of(escx_handle_s* h, int L) { return 1 + L / h->cfg.hop_length; }
// frame count to reserve when only the latent width is known (decode first): the largest T that maps to W
static int frames_for_width(escx_handle_s* h, int W) {
    return h->cfg.patch_t * W + h->cfg.patch_t - 1;
}

namespace {
struct TmpBuf {                      // test-path scratch for the stage-level entry points (synchronous)
    float* p = nullptr;
    ~TmpBuf() { if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); } }
    int alloc(size_t n_floats) { return hipMalloc((void**)&p, std::max<size_t>(n_floats, 1) * sizeof(float)) == hipSuccess ? 0 : -1; }
};
}  // namespace

// Fused MLP on the bf16 matrix cores with every fp32 operand split exactly into three bf16 terms (fused_mlp_x3.h): DEFAULT for every instantiated width
// (48, 80, 96, 144, 192, 384).  ESCX_MLP_X3=<max padded width> restricts it, ESCX_MLP_X3=0 = the fp32-MFMA kernel (fused_mlp.h) everywhere - the
// round-4 arithmetic, which bench.py also reports (`fp32_mfma_only`) and tests/test_gpu_parity.py keeps as an arm.
// Terms per operand of the split-operand kernels (split_terms.h): 2 (default) = two fp16 terms + power-of-two weight scales, three cross products;
// ESCX_X3_TERMS=3 = three bf16 terms, exact split, six cross products (the first round-5 form).  One switch for the MLPs, the Q / K / V projections and PatchMerge / PatchSplit.
// (round 6: the mode is a field of the handle, escx_set_precision; escx_handle_s::prec)
static int x3_nt(const escx_handle_s* h) { return h->prec == 3 ? 3 : 2; }        // terms per operand of the split images / kernels (meaningful when prec != 0)

// Waves per workgroup (4 or 8) for the fused kernels.  A wave owns `units` 16-row tiles; a workgroup's waves spread over the
// 4 SIMDs of a CU and workgroups are dealt round-robin to the 256 CUs, so the makespan in tile-times is
//   ceil(workgroups / 256) * ceil(NW / 4) * units.
// 8 waves share each weight fragment between twice as many rows (half the L2->LDS traffic) and win ties.
// `cap` = waves per SIMD the kernel's register budget allows (mlp_min_waves / attn_min_waves): at 3, an 8-wave workgroup leaves
// the third slot of every SIMD empty (one workgroup = 2 waves per SIMD, a second one does not fit), so 4-wave workgroups it is.
// Waves per workgroup of the split-operand MLP (fused_mlp_x3.h), measured at 36 clips (tools/ab.py, ESCX_MLP_X3_NW in tagged builds; ms per step alone, 4 -> 8 waves):
// C = 45 1.25 -> 1.17, C = 72 0.96 -> 0.84, C = 96 1.13 -> 0.97, C = 144 1.34 -> 1.55, C = 192 0.97 -> 1.00, C = 384 1.59 -> 1.11 (8 waves cap the kernel at 256
// registers with a small spill, but put two waves on every SIMD).  Small grids keep the fp32 kernel's rule (`variant`): a wave per SIMD first.
static int mlp_x3_nw(int M, int Cp, int variant) {
    const long long tiles = (M + 15) / 16;
    if (Cp == 144 || Cp == 192) return 4;
    if (Cp == 384) return variant == 3 ? 8 : 4;
    return tiles >= 8 * 512 ? 8 : (variant == 3 ? 8 : 4);
}

static int pick_nw(long long tiles, int units, int cap = 4) {
    static const bool cap_rule = [] { const char* e = ESCX_TUNE_ENV("ESCX_NW_CAP_RULE"); return !(e && e[0] == '0'); }();
    if (cap == 3 && cap_rule) return 4;
    auto cost = [&](int nw) { const long long wg = (tiles / units + nw - 1) / nw; return ((wg + 255) / 256) * ((nw + 3) / 4) * units; };
    return cost(8) <= cost(4) ? 8 : 4;
}
static int mlp_cap(int Cp) { return Cp <= ESCX_MLP_OCC4 ? 4 : (Cp <= 192 ? 3 : 1); }                              // fused_mlp.h: mlp_min_waves<CP, 1>
static int attn_cap(int Cp, int tmw) { const int v = Cp * tmw; return v <= 96 ? 4 : (v <= 160 ? 3 : (v <= 192 ? 2 : 1)); }   // fused_attn.h: attn_min_waves
// Hidden split of the fused MLP (fused_mlp.h): at the deep scales a clip has so few token rows (600 / 1200 at C = 384 / 192
// for 3 s) that one wave per 16-row tile cannot fill 1024 SIMDs at serving batch sizes, so three workgroups share a row block
// and a combine pass adds their fc2 partial sums in fixed order.  The rule depends on the layer geometry only - never on the
// batch - so that a clip's arithmetic (and therefore its codes) is identical in any batch or shard it is processed in.
// The partial sums live in the (otherwise unused) hidden-activation buffer of the unfused path: needs hiddenP >= 3 * Cp.
static int mlp_hs_for(int tokens_per_clip, int HT, int Cp) { static const int lim = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_HS_TOKENS"); return e && e[0] ? atoi(e) : 1200; }(); return (tokens_per_clip <= lim && HT % 3 == 0 && HT * 16 >= 3 * Cp) ? 3 : 1; }
// The same split over the head groups of the fused attention (partials share the buffer; an MLP always follows on the same
// stream).  Like the hidden split it would have to depend on the clip's geometry only, never on the batch (it re-associates the projection
// sum).  Measured with ESCX_ATTN_GS_TOKENS=600 (the C = 384 scale of a 3 s clip; tools/small_batch.py): one clip 3.52 -> 3.12 ms, 4 clips
// 4.30 -> 3.90, 8 clips 5.56 -> 5.19 ms, but 36 clips +0.2 .. 0.8 % and 288 clips 121.4 -> 123.5 ms in same-box A/B runs (a wave holds one window pair and is
// MFMA-bound on its own: three workgroups per pair put a small grid on three times the CUs, a full grid gains nothing and pays the
// combine).  The throughput configurations are the ones BASELINE quotes, so the split is OFF by default; all parity tests and the
// 576-clip sweep are bit-exact with it on, a latency-bound deployment can switch it on.
// Round 4: decided on parity evidence, not on the 36-clip throughput.  Oracle sweeps of BOTH settings on the same clips
// (profiles/r4_parity_sweep_*_gs_{on,off}.log): ESC-Base 576 clips - split ON 576 / 576 bit-exact, split OFF 575 / 576 (one near-tie code); ESC-Large 288
// clips - 286 / 288 either way (the same two near-tie clips).  The split is therefore ON for every batch size (geometry rule: maps of up to 600
// tokens per clip = the C = 384 scale of a 3 s clip).  Cost / gain on the day's build (tools/ab.py): 36 clips 15.79 -> 15.87 ms (+0.5 %), 8 clips
// 5.59 -> 5.18 ms, one clip 3.52 -> 3.12 ms.  ESCX_ATTN_GS_TOKENS=0 switches it off.
static int attn_gs_for(const escx_handle_s* h, int tokens_per_clip, int n_groups, int hiddenP, int Cp) { return (tokens_per_clip <= h->attn_gs_tokens && n_groups % 3 == 0 && hiddenP >= 3 * Cp) ? 3 : 1; }
static int mlp_variant_for(int M, int Cp) { return pick_nw((M + 15) / 16, 1, mlp_cap(Cp)) == 8 ? 3 : 1; }     // fused_swin.hip: 1 = (TM 1, NW 4), 3 = (TM 1, NW 8)
// The hidden-split MLP at C >= 384: 8-wave workgroups when that fills one dispatch round anyway (36-clip batches: 255 workgroups, half the
// weight DMA per wave), 4-wave ones for small grids - with 8 waves two waves share every SIMD's MFMA pipe and a 15-workgroup launch takes
// as long as a 255-workgroup one (single clip: 197 us; 4 waves: one wave per SIMD).  A wave's rows and arithmetic are the same either way.
// Measured (tools/small_batch.py, ms per encode+decode, 8-wave -> rule): B = 1 3.88 -> 3.52, B = 4 4.67 -> 4.30, B = 8 5.96 -> 5.60, B = 12 6.79 -> 6.64;
// at 8 clips per part (225 four-wave workgroups) the two forms cross over.
static int mlp_split_nw(int M, int hs) {
    const long long tiles = (M + 15) / 16;
    return ((tiles + 3) / 4) * hs <= 192 ? 4 : 8;
}

// One TransformerLayer on padded token maps.  x_in is read-only; y receives (B, H'*W, CoutP).
// attention.py:48-91 (layer), 129-178 (block): LN1 -> pad -> roll -> windows -> attention -> reverse -> residual -> MLP.
static int run_layer(escx_handle_s* h, const Layer& L, const float* x_in, float* y, int B, int H, int W, int* Hout, hipStream_t st) {
    const int tokens = H * W, M = B * tokens;
    const int ws = h->ws, NW2 = ws * ws;
    const int Hp = rup(H, ws), Wp = rup(W, ws), slots = Hp * Wp, Ms = B * slots;
    float* cur = L.scale ? h->work : y;
    const float* src = x_in;
    int rc;
    // Combine of a hidden-split MLP (fused_mlp.h), alternative form: not a launch of its own - the NEXT kernel of the layer (the following block's
    // fused attention, or the merge / split) forms x + (((P0 + P1) + P2) + bias) while it loads its rows (same arithmetic, same order: bit-identical).
    // Motivation: in the two-stream execution the 20 short combine launches per step cost 2.6x their isolated time (they queue behind the other batch
    // part's resident workgroups).  Consumers without a combine-on-load instantiation get the explicit rows_combine launch.
    // OPT-IN (ESCX_COMBINE_ON_LOAD=1).  MEASURED (round 4, B = 36, profiles/r4_mlp_combine_ab.txt): bit-identical, and slower - the consumers' gather
    // prologues are latency-bound and badly coalesced (64-byte pieces), four row reads there (twice in the attention: LayerNorm input and shortcut) cost
    // more than the streaming combine launch at 6 TB/s: attention C = 192 / 384 +0.14 / +0.13, merge / split +0.18 ms per step against 0.30 removed.
    static const bool comb_on_load = [] { const char* e = ESCX_TUNE_ENV("ESCX_COMBINE_ON_LOAD"); return e && e[0] == '1'; }();
    CombineOnLoad pend{nullptr, nullptr, 0, 0};
    std::string pend_tag;
    auto flush_pending = [&]() {                        // explicit combine launch (fallback)
        if (pend.n > 0) {
            const double dMp = M;
            PROF("mlp_combine" + pend_tag, 0, (pend.n + 2) * dMp * L.Cp * sizeof(float), rows_combine(cur, cur, pend.partial, pend.bias, M, L.Cp, pend.n, st));
            pend.n = 0;
        }
    };
    for (size_t j = 0; j < L.blocks.size(); ++j) {
        const BlockW& bw = L.blocks[j];
        const int shift = (j % 2 == 0) ? 0 : ws / 2;                          // attention.py:29
        const int* map;
        if ((rc = get_map(h, H, W, shift, &map))) return rc;
        const std::string tag = h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string();
        const double dM = M, dMs = Ms, dC = L.C, f4 = sizeof(float);
        bool attn_done = false;
        if (h->use_fused && h->use_fused_attn && L.attn_mode >= 0) {
            int frc = 0;
            int nw = h->attn_nw ? h->attn_nw : ((L.Cp > 192 || (L.Cp == 192 && bw.x3a)) ? 4 : pick_nw(Ms / 16, attn_windows_per_wave(L.Cp), attn_cap(L.Cp, attn_windows_per_wave(L.Cp))));    // 8 waves cap the kernel at 256 VGPRs: spills above C = 192
            if (H == 2 && W % 4 == 0 && h->attn_pack) nw = -(h->attn_nw ? h->attn_nw : (L.Cp > 192 ? 4 : pick_nw((Ms / 16 + 1) / 2, 1)));    // packed half-window pairs
            const double proj_rows = nw < 0 ? dM : dMs;         // packed pairs project only the real tokens
            int gs = h->attn_gs > 0 ? (L.hiddenP >= h->attn_gs * L.Cp ? h->attn_gs : 1) : attn_gs_for(h, tokens, L.n_groups, L.hiddenP, L.Cp);
            bool launched = false;
            if (pend.n > 0) {                           // block input still split over the previous MLP's slabs: combine on load if this kernel can
                int gs0 = gs;
                const size_t n_recs = h->prof_recs.size();
                PROF("attn_fused" + tag, 2 * proj_rows * dC * 4 * dC + 4 * dMs * 16 * dC, (2 + pend.n) * dM * dC * f4,
                     frc = attn_fused(src, cur, L.Cp, L.C, L.attn_mode, L.n_groups, bw.ln1_g, bw.ln1_b, bw.waf, bw.baf, bw.bias_tab_f, bw.bproj,
                                      map, slots, tokens, Ms / 16, Hp / 4, Wp / 4, shift > 0, 1.0f / std::sqrt((float)L.hd), nw, &gs0, h->hid, M, st, &pend));
                if (frc == 0) { pend.n = 0; gs = gs0; launched = true; }
                else { if (h->prof_recs.size() > n_recs) h->prof_recs.pop_back(); flush_pending(); }      // nothing was launched: no record, explicit combine
            }
            if (!launched)
            PROF("attn_fused" + tag, 2 * proj_rows * dC * 4 * dC + 4 * dMs * 16 * dC, 2 * dM * dC * f4,
                 frc = attn_fused(src, cur, L.Cp, L.C, L.attn_mode, L.n_groups, bw.ln1_g, bw.ln1_b, bw.waf, bw.baf, bw.bias_tab_f, bw.bproj,
                                  map, slots, tokens, Ms / 16, Hp / 4, Wp / 4, shift > 0, 1.0f / std::sqrt((float)L.hd), nw, &gs, h->hid, M, st, nullptr, nullptr, (nw > 0 || L.Cp == 384) ? bw.x3a : nullptr, bw.x3a_pairs ? 1 : (x3_nt(h) == 2 ? 2 : 0)));      // C = 384: the split stream exists for the packed (nw < 0) kernel only
            attn_done = (frc == 0);
            if (attn_done && gs > 1)
                PROF("attn_combine" + tag, 0, (gs + 2) * dM * L.Cp * f4, rows_combine(cur, src, h->hid, bw.bproj, M, L.Cp, gs, st));
        }
        if (!attn_done) {
        flush_pending();
        PROF("ln1_gather" + tag, 0, (dM + dMs) * dC * f4,
             ln_rows(1, src, h->xn, bw.ln1_g, bw.ln1_b, map, slots, tokens, Ms, L.C, L.Cp, st));
        PROF("gemm_qkv" + tag, 2 * dMs * dC * 3 * dC, (dMs * 4 * dC + 3 * dC * dC) * f4,
             gemm_qkv(h->xn, L.Cp, Ms, bw.wqkv, L.Nqkv, L.Cp, h->qkv, bw.bqkv, L.nH * L.hdp, 1.0f / std::sqrt((float)L.hd), st));
        int arc = 0;
        if (ws != 4)
        PROF("window_attn_any" + tag, 4 * dMs * NW2 * dC, dMs * 4 * dC * f4,
             arc = window_attention_any(h->qkv, bw.bias_tab, h->obuf, Ms / NW2, ws, L.nH, L.hd, L.hdp, L.Nqkv, L.Ko, Hp / ws, Wp / ws, shift, st));
        else
        PROF("window_attn" + tag, 4 * dMs * 16 * dC, dMs * 4 * dC * f4,
             arc = window_attention(h->qkv, bw.bias_tab, h->obuf, Ms / 16, L.nH, L.hdp, L.Nqkv, L.Ko, Hp / 4, Wp / 4, shift > 0, st));
        if (arc) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "head_dim %d unsupported by the attention kernel", L.hd);
        PROF("gemm_proj" + tag, 2 * dMs * dC * dC, (dMs * dC + 2 * dM * dC + dC * dC) * f4,
             gemm_proj_scatter(h->obuf, L.Ko, Ms, bw.wproj, L.Cp, L.Ko, cur, src, bw.bproj, map, slots, tokens, st));
        }
        if (h->use_fused) {
            int frc = 0;
            static const int hs_nw8_cp = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_HS_NW8_CP"); return e && e[0] ? atoi(e) : 384; }();
            static const int tm2_max = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_TM2_MAXCP"); return e && e[0] ? atoi(e) : 0; }();
            static const int tm2_nw8 = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_TM2_NW8"); return e && e[0] == '1'; }();
            bool combined = false;
            int hs = (h->mlp_hs > 0 && L.hiddenP >= h->mlp_hs * L.Cp) ? h->mlp_hs : mlp_hs_for(tokens, L.hiddenP / 16, L.Cp);
            const int variant = h->mlp_variant >= 0 ? h->mlp_variant : (hs > 1 ? ((L.Cp >= hs_nw8_cp && mlp_split_nw(M, hs) == 8) ? 3 : 1) : (L.Cp <= tm2_max ? (tm2_nw8 ? 5 : 4) : mlp_variant_for(M, L.Cp)));
            static const int x3_nw_force = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLP_X3_NW"); return e && e[0] ? atoi(e) : 0; }();
            if (bw.x3w && pend.n == 0) {       // three-term bf16 split on the bf16 matrix cores (fused_mlp_x3.h); same hidden-split rule and combine
                if (h->mlp_split_fold && L.scale == 2 && j + 1 == L.blocks.size() && hs == 1 && L.sub_x3s) {      // PatchSplit in the epilogue of the layer's last MLP
                    const MlpSplit sp{reinterpret_cast<const float*>(L.sub_x3s), L.sub_g, L.sub_b, y, 2 * L.CoutP / 16, H, W, L.CoutP};
                    int src3 = -1, one = 1;
                    PROF("mlp_x3_split" + tag, 4 * dM * dC * L.hidden + 2.0 * dM * dC * 2 * L.Cout, (dM * dC + dM * 2 * L.Cout) * f4,
                         src3 = mlp_x3(cur, M, L.C, L.Cp, bw.ln2_g, bw.ln2_b, bw.b1, bw.b2, bw.x3w, L.hiddenP, x3_nw_force ? x3_nw_force : mlp_x3_nw(M, L.Cp, variant), &one, nullptr, st, &sp, x3_nt(h)));
                    if (src3 == 0) { *Hout = 2 * H; return launch_ok(L.prefix.c_str()); }
                    if (h->prof && !h->prof_recs.empty()) h->prof_recs.pop_back();
                }
                int xrc = -1, xhs = hs;
                PROF("mlp_x3" + tag, 4 * dM * dC * L.hidden, 2 * dM * dC * f4,
                     xrc = mlp_x3(cur, M, L.C, L.Cp, bw.ln2_g, bw.ln2_b, bw.b1, bw.b2, bw.x3w, L.hiddenP, x3_nw_force ? x3_nw_force : mlp_x3_nw(M, L.Cp, variant), &xhs, h->hid, st, nullptr, x3_nt(h)));
                if (xrc == 0) {
                    if (xhs > 1) { pend = CombineOnLoad{h->hid, bw.b2, (long long)M * L.Cp, xhs}; pend_tag = tag; flush_pending(); }
                    src = cur; continue;
                }
                if (h->prof && !h->prof_recs.empty()) h->prof_recs.pop_back();
            }
            // PatchSplit in the epilogue of the layer's last MLP (fused_mlp.h SPLIT): one launch and one HBM round trip of the
            // pre-split map less.  ESCX_MLP_SPLIT_FOLD=0: the separate LN + linear launch (rowgemm_fused_kernel), as before.
            if (h->mlp_split_fold && L.scale == 2 && j + 1 == L.blocks.size() && hs == 1 && pend.n == 0) {
                const MlpSplit sp{L.sub_wf, L.sub_g, L.sub_b, y, 2 * L.CoutP / 16, H, W, L.CoutP};
                int sfrc = -1;
                PROF("mlp_split_fused" + tag, 4 * dM * dC * L.hidden + 2.0 * dM * dC * 2 * L.Cout, (dM * dC + dM * 2 * L.Cout) * f4,
                     sfrc = mlp_fused(cur, M, L.C, L.Cp, bw.ln2_g, bw.ln2_b, bw.w1f, bw.b1, bw.w2f, bw.b2, bw.wcf, L.hiddenP, variant, &hs, h->hid, st, nullptr,
                                     nullptr, 0, nullptr, &sp));
                if (sfrc == 0) { *Hout = 2 * H; return launch_ok(L.prefix.c_str()); }
                if (h->prof && !h->prof_recs.empty()) h->prof_recs.pop_back();      // no SPLIT instantiation for this width: nothing was launched
            }
            PROF("mlp_fused" + tag, 4 * dM * dC * L.hidden, 2 * dM * dC * f4,
                 frc = mlp_fused(cur, M, L.C, L.Cp, bw.ln2_g, bw.ln2_b, bw.w1f, bw.b1, bw.w2f, bw.b2, bw.wcf, L.hiddenP, variant, &hs, h->hid, st, nullptr,
                                 h->tickets, WsFields::N_TICKETS, &combined));
            if (frc == 0 && hs > 1 && !combined) {
                pend = CombineOnLoad{h->hid, bw.b2, (long long)M * L.Cp, hs}; pend_tag = tag;
                if (!comb_on_load) flush_pending();
            }
            if (frc == 0) { src = cur; continue; }
        }
        PROF("ln2" + tag, 0, 2 * dM * dC * f4,
             ln_rows(0, cur, h->xn, bw.ln2_g, bw.ln2_b, nullptr, tokens, tokens, M, L.C, L.Cp, st));
        PROF("gemm_fc1_gelu" + tag, 2 * dM * dC * L.hidden, (dM * (dC + L.hidden) + dC * L.hidden) * f4,
             gemm_gelu(h->xn, L.Cp, M, bw.w1, L.hiddenP, L.Cp, h->hid, bw.b1, st));
        PROF("gemm_fc2_res" + tag, 2 * dM * dC * L.hidden, (dM * (2 * dC + L.hidden) + dC * L.hidden) * f4,
             gemm_residual(h->hid, L.hiddenP, M, bw.w2, L.Cp, L.hiddenP, cur, bw.b2, cur, st));
        src = cur;
    }
    if (L.scale == 1) {
        const int H2 = (H + 1) / 2;
        const int* map;
        if ((rc = get_map(h, H, W, -1, &map))) return rc;
        int mrc = -1;
        if (h->use_fused && pend.n > 0) {
            const size_t n_recs = h->prof_recs.size();
            PROF("merge_fused" + (h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string()), 2.0 * B * H2 * W * 2 * L.C * L.Cout, ((double)M * L.C * (1 + pend.n) + (double)B * H2 * W * L.Cout) * 4,
                 mrc = rowgemm_fused(2, cur, y, L.sub_g, L.sub_b, L.sub_wf, map, B * H2 * W, H2 * W, tokens, L.C, L.Cp, L.CoutP, 0, 0, 0, 0, st, &pend));
            if (mrc == 0) pend.n = 0; else if (h->prof_recs.size() > n_recs) h->prof_recs.pop_back();
        }
        flush_pending();
        if (h->use_fused && mrc != 0)
            PROF("merge_fused" + (h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string()), 2.0 * B * H2 * W * 2 * L.C * L.Cout, ((double)M * L.C + (double)B * H2 * W * L.Cout) * 4,
                 mrc = rowgemm_fused(2, cur, y, L.sub_g, L.sub_b, L.sub_wf, map, B * H2 * W, H2 * W, tokens, L.C, L.Cp, L.CoutP, 0, 0, 0, 0, st, nullptr, L.sub_x3, x3_nt(h)));
        if (mrc != 0) {
        PROF("merge_ln", 0, 2.0 * M * L.C * 4,
             ln_rows(2, cur, h->xn, L.sub_g, L.sub_b, map, H2 * W, tokens, B * H2 * W, L.C, L.Cp, st));
        PROF("merge_gemm", 2.0 * B * H2 * W * 2 * L.C * L.Cout, (double)B * H2 * W * (2 * L.C + L.Cout) * 4,
             gemm_store(h->xn, 2 * L.Cp, B * H2 * W, L.sub_w, L.CoutP, 2 * L.Cp, y, L.CoutP, nullptr, st));
        }
        *Hout = H2;
    } else if (L.scale == 2) {
        int src2 = -1;
        if (h->use_fused && pend.n > 0) {
            const size_t n_recs = h->prof_recs.size();
            PROF("split_fused" + (h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string()), 2.0 * M * L.C * 2 * L.Cout, (double)M * (L.C * (1 + pend.n) + 2 * L.Cout) * 4,
                 src2 = rowgemm_fused(1, cur, y, L.sub_g, L.sub_b, L.sub_wf, nullptr, M, tokens, tokens, L.C, L.Cp, 2 * L.CoutP, 1, H, W, L.CoutP, st, &pend));
            if (src2 == 0) pend.n = 0; else if (h->prof_recs.size() > n_recs) h->prof_recs.pop_back();
        }
        flush_pending();
        if (h->use_fused && src2 != 0)
            PROF("split_fused" + (h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string()), 2.0 * M * L.C * 2 * L.Cout, (double)M * (L.C + 2 * L.Cout) * 4,
                 src2 = rowgemm_fused(1, cur, y, L.sub_g, L.sub_b, L.sub_wf, nullptr, M, tokens, tokens, L.C, L.Cp, 2 * L.CoutP, 1, H, W, L.CoutP, st, nullptr, L.sub_x3, x3_nt(h)));
        if (src2 != 0) {
        PROF("split_ln", 0, 2.0 * M * L.C * 4,
             ln_rows(0, cur, h->xn, L.sub_g, L.sub_b, nullptr, tokens, tokens, M, L.C, L.Cp, st));
        PROF("split_gemm", 2.0 * M * L.C * 2 * L.Cout, (double)M * (L.C + 2 * L.Cout) * 4,
             gemm_split(h->xn, L.Cp, M, L.sub_w, 2 * L.CoutP, L.Cp, y, H, W, L.CoutP, st));
        }
        *Hout = 2 * H;
    } else {
        flush_pending();
        *Hout = H;
    }
    return launch_ok(L.prefix.c_str());
}

static int run_stft(escx_handle_s* h, const float* wave, int B, int L, int T, float* spec, hipStream_t st) {
    PROF("stft_dft_gemm", 2.0 * B * T * h->cfg.win_length * 2 * h->F, ((double)B * L + (double)B * T * 2 * h->F) * 4,
         gemm_frames(wave, B, L, T, h->cfg.hop_length, h->left - h->n_fft / 2, h->dft_w, h->cfg.in_dim * h->Fp, h->winP, spec, st));
    return launch_ok("stft");
}

static int run_patch_embed(escx_handle_s* h, const float* spec, int B, int T, int W, float* tok, hipStream_t st) {
    const escx_config& c = h->cfg;
    const int H0 = c.in_freq / c.patch_f;
    const double toks = (double)B * H0 * W;
    PROF("patch_embed_gemm", 2.0 * toks * c.in_dim * c.patch_f * c.patch_t * h->C0, ((double)B * T * 2 * h->F + toks * h->C0) * 4,
         gemm_patch(spec, B, T, c.in_dim, h->Fp, H0, W, c.patch_f, c.patch_t, h->pe_w, h->C0p, h->Kpe, tok, h->pe_b, st));
    PROF("patch_embed_ln", 0, 2 * toks * h->C0 * 4,
         ln_rows(0, tok, tok, h->pe_g, h->pe_beta, nullptr, H0 * W, H0 * W, B * H0 * W, h->C0, h->C0p, st));
    return launch_ok("patch_embed");
}

static int run_encoder(escx_handle_s* h, const Shapes& s, hipStream_t st) {     // base.py:143-158
    int rc, H = s.H0, Hn;
    if ((rc = run_patch_embed(h, h->spec, s.B, s.T, s.W, h->stageA, st))) return rc;
    if ((rc = run_layer(h, h->layers[0], h->stageA, h->enc_hs[0], s.B, H, s.W, &Hn, st))) return rc;
    for (int i = 0; i + 1 < h->n; ++i) {
        if ((rc = run_layer(h, h->layers[1 + i], h->enc_hs[i], h->enc_hs[i + 1], s.B, H, s.W, &Hn, st))) return rc;
        H = Hn;
    }
    return 0;
}

// De-quantisation tables of the product quantisers (Quant::tab, escx_internal.h): out = dec + up_proj_g(codebook_g[code]) becomes a table-row add.
// Built by pvq_up_kernel ITSELF on Ksz pseudo-vectors (vector k carries code k in every group; one "clip" of ov * Ksz frames), so a table entry is
// bit for bit what the up-projection MFMA chain produces for that code (inference only - like the folded de-embedding the tables are derived
// state: finalisation and every device-side parameter refresh mark them stale, the next inference entry rebuilds them on the caller's stream
// before the batch parts fork).  ESCX_PVQ_TABLE=0: no tables (up-projection on the MFMA, the round-4 form).
static int ensure_pvq_tables(escx_handle_s* h, hipStream_t st) {
    if (!h->pvq_tab_stale) return 0;
    const escx_config& c = h->cfg;
    const int G = c.group_size, Ksz = c.codebook_size;
    const bool on = h->pvq_table;
    // Derived buffers are allocated the first time they are wanted and KEPT (a precision switch or a parameter refresh only re-launches the pack kernels: no hipFree,
    // no device synchronisation on the way - ADVICE r5); the ACTIVE pointer says whether the launch sequence uses them.
    auto grab = [&](void** buf, size_t bytes) -> bool { return *buf || hipMalloc(buf, bytes) == hipSuccess; };
    if (on && !h->iota_codes) {
        std::vector<long long> iota((size_t)G * Ksz);
        for (int g = 0; g < G; ++g) for (int k = 0; k < Ksz; ++k) iota[(size_t)g * Ksz + k] = k;
        ESCX_HIP(hipMalloc((void**)&h->iota_codes, iota.size() * sizeof(long long)));
        ESCX_HIP(hipMemcpy(h->iota_codes, iota.data(), iota.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    for (Quant& q : h->quants) {
        if (!on || !q.tab_ok) { if (q.tab) { ESCX_HIP(hipDeviceSynchronize()); (void)hipFree(q.tab); q.tab = nullptr; } continue; }
        if (!q.tab) ESCX_HIP(hipMalloc((void**)&q.tab, (size_t)Ksz * q.Kq * sizeof(float)));
        if (pvq_up(h->iota_codes, (long long)G * Ksz, q.cbraw, G, Ksz, q.dt, 1, q.Hq, c.overlap * Ksz, q.Cp, c.overlap, q.wup, q.Kq, q.Kup, nullptr, q.tab, st) != 0) {
            ESCX_HIP(hipDeviceSynchronize()); (void)hipFree(q.tab); q.tab = nullptr;      // no up-projection kernel for this width: the engine form stays
        }
    }
    const int nt = x3_nt(h);
    const bool split_on = h->prec != 0;
    // tagged builds only: the two-term images WITHOUT the activation scales of the range rule (what tests/test_gpu_parity.py test_range_stress_checkpoints guards against)
    static const bool no_act_scale = [] { const char* e = ESCX_TUNE_ENV("ESCX_X2_NO_ACT_SCALE"); return e && e[0] == '1'; }();
    const int x3_max = split_on ? h->mlp_x3_max : 0;
    // the two-term fp16 stream of the halo-tiled de-embedding (fused_deembed.h deembed7_x2_kernel): in the two-term mode only - the three-term mode and the fp32-MFMA
    // mode keep the fp32 kernel (the latter stays bit-identical to round 4)
    {
        const bool want = x3_max > 0 && h->prec == 2 && h->deembed_halo && h->dch_w && deembed7_x2_image_bytes(h->C0p) > 0;
        h->dch_x2 = nullptr;
        if (want && grab(&h->dch_x2_buf, deembed7_x2_image_bytes(h->C0p)) && deembed7_x2_pack(h->dch_w, h->dch_x2_buf, h->C0p, st) == 0) h->dch_x2 = h->dch_x2_buf;
    }
    // the split weight images of the fused MLP (fused_mlp_x3.h)
    for (Layer& L : h->layers)
        for (BlockW& bw : L.blocks) {
            const bool want = x3_max > 0 && L.Cp <= x3_max && (L.Cp == 48 || L.Cp == 80 || L.Cp == 96 || L.Cp == 144 || L.Cp == 192 || L.Cp == 384) && L.hiddenP % 32 == 0;
            bw.x3w = nullptr;
            if (want && grab(&bw.x3w_buf, mlp_x3_bytes(L.Cp, L.hiddenP, 3)) &&         // sized for the larger (three-term) image
                mlp_x3_pack(bw.w1, bw.w2, bw.x3w_buf, L.Cp, L.hiddenP, st, nt, no_act_scale ? nullptr : bw.ln2_g, bw.ln2_b, bw.b1, L.C) == 0) bw.x3w = bw.x3w_buf;
        }
    // the split Q / K / V weight streams of the fused attention (fused_attn.h X3)
    const int ax3_max = split_on ? h->attn_x3_max : 0;
    for (Layer& L : h->layers)
        for (BlockW& bw : L.blocks) {
            const bool want = ax3_max > 0 && L.Cp <= ax3_max && L.attn_mode >= 0 && h->use_fused_attn &&
                              ((L.Cp == 48 && L.attn_mode == 0) || (L.Cp == 80 && L.attn_mode != 1) || (L.Cp == 96 && L.attn_mode != 2) || (L.Cp == 144 && L.attn_mode != 2) || (L.Cp == 192 && L.attn_mode == 1) || (L.Cp == 384 && L.attn_mode == 0));      // 384: the packed H = 2 kernel only (the launcher falls back to fp32 elsewhere)
            bw.x3a = nullptr;
            if (!want || !grab(&bw.x3a_buf, attn_x3_bytes(L.Cp, L.attn_mode, L.n_groups))) continue;
            // pair order (the output projection in split form too) where two head groups always travel together: mode 0 / 1, group counts that stay even under the 3-way head-group split
            // OPT-IN, tagged builds (ESCX_ATTN_X3_PAIRS=1): measured no faster - the operand split of the O^T tiles and the two extra stage barriers per pair eat the
            // matrix time saved, and at C = 192 the kernel drops to one wave per SIMD (profiles/r5_attn_ab.txt)
            static const bool pairs_on = [] { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_X3_PAIRS"); return e && e[0] == '1'; }();
            bw.x3a_pairs = pairs_on && nt == 3 && L.attn_mode != 2 && L.Cp != 48 && L.n_groups % 2 == 0 && (L.n_groups % 3 != 0 || (L.n_groups / 3) % 2 == 0);
            if (attn_x3_pack(bw.waf, bw.x3a_buf, L.Cp, L.attn_mode, L.n_groups, st, bw.x3a_pairs ? 1 : (nt == 2 ? 2 : 0), no_act_scale ? nullptr : bw.ln1_g, bw.ln1_b, L.C, bw.baf) == 0) bw.x3a = bw.x3a_buf;
        }
    // the split weight streams of PatchMerge / PatchSplit (fused_rowgemm.h rowgemm_x3_kernel)
    for (Layer& L : h->layers) {
        const int KP = L.scale == 1 ? 2 * L.Cp : L.Cp, Np = L.scale == 1 ? L.CoutP : 2 * L.CoutP;
        const bool want = split_on && h->rowgemm_x3 && L.scale != 0 && L.sub_wf && (KP == 80 || KP == 96 || KP == 144 || KP == 160 || KP == 192 || KP == 288 || KP == 384);
        L.sub_x3 = nullptr; L.sub_x3s = nullptr;
        if (!want || !grab(&L.sub_x3_buf, rowgemm_x3_bytes(KP, Np))) continue;
        if (rowgemm_x3_pack(L.sub_wf, L.sub_x3_buf, KP, Np, st, nt, no_act_scale ? nullptr : L.sub_g, L.sub_b, (L.scale == 1 ? 2 : 1) * L.C) == 0) L.sub_x3 = L.sub_x3_buf;
        if (L.scale == 2 && (L.Cp == 80 || L.Cp == 96 || L.Cp == 144) && L.Cp <= x3_max && grab(&L.sub_x3s_buf, mlp_x3_split_bytes(L.Cp, Np)) &&       // PatchSplit folded into the split-operand MLP's epilogue
            mlp_x3_split_pack(L.sub_wf, L.sub_x3s_buf, L.Cp, Np, st) == 0) L.sub_x3s = L.sub_x3s_buf;
    }
    h->pvq_tab_stale = false;
    return launch_ok("derived images");
}

// escx_set_precision / escx_get_precision (include/escx.h): the arithmetic of the K = C contractions of this handle.  Takes effect with the next call: the
// split images are derived state and are re-packed on that call's stream.
extern "C" int escx_set_precision(escx_handle h, int mode) {
    if (!h) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    if (mode != ESCX_PRECISION_FP32 && mode != ESCX_PRECISION_BF16X3 && mode != ESCX_PRECISION_F16X2)
        ESCX_FAIL(ESCX_ERR_INVALID_ARG, "precision mode %d (0 = fp32 MFMA, 3 = three bf16 terms, 2 = two fp16 terms)", mode);
    if (mode != h->prec) { h->prec = mode; h->pvq_tab_stale = true; }
    return ESCX_OK;
}
extern "C" int escx_get_precision(escx_handle h) { return h ? h->prec : ESCX_ERR_INVALID_ARG; }

static int run_pvq_encode(escx_handle_s* h, const Quant& q, const float* enc, const float* dec, int B, int W, long long* codes,
                          long long bstride, float* loss, hipStream_t st) {
    const escx_config& c = h->cfg;
    const int Tq = W / c.overlap, M = B * Tq;
    const int splits = pvq_down_splits(M, q.Kq, q.Cp);
    if ((size_t)splits * M * q.Nz > h->zpart_cap) ESCX_FAIL(ESCX_ERR_STATE, "split-K scratch too small");
    const double vec = (double)c.overlap * q.Hq * q.C;
    static const bool special = [] { const char* e = ESCX_TUNE_ENV("ESCX_PVQ_DOWN_KERNEL"); return e && e[0] == '1'; }();   // opt-in: bit-identical but slower than the engine form (profiles/r4_pvq_ab.txt)
    int drc = -1;
    if (special)
        PROF("pvq_down_gemm", 2.0 * M * vec * q.d, (double)M * vec * (dec ? 2 : 1) * 4,
             drc = pvq_down(enc, dec, B, q.Hq, W, q.Cp, c.overlap, q.wd, q.Nz, q.Kq, h->zpart, splits, pvq_down_bk(q.Cp), st));
    if (drc != 0) {
        if (special && h->prof && !h->prof_recs.empty()) h->prof_recs.pop_back();
    PROF("pvq_down_gemm", 2.0 * M * vec * q.d, (double)M * vec * (dec ? 2 : 1) * 4,
         gemm_pvq_down(enc, dec, B, q.Hq, W, q.Cp, c.overlap, q.wd, q.Nz, q.Kq, h->zpart, splits, st));
    }
    int src_rc = 0;
    PROF("pvq_search", 2.0 * M * c.group_size * c.codebook_size * q.d, ((double)c.group_size * c.codebook_size * q.d + (double)M * c.group_size * q.d) * 4,
         src_rc = pvq_search(h->zpart, splits, M, q.Nz, q.cbn, q.c2, q.cbraw, c.group_size, c.codebook_size, q.d, q.dt, Tq, codes, bstride,
                             loss, 1.0f / ((float)Tq * q.d * c.group_size), c.l2norm, st));
    if (src_rc) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "codebook_dim %d unsupported by the search kernel", q.d);
    return launch_ok("pvq_encode");
}

static int run_pvq_decode(escx_handle_s* h, const Quant& q, const long long* codes, long long bstride, const float* dec, int B, int W,
                          float* out, hipStream_t st) {
    const escx_config& c = h->cfg;
    const double vec = (double)c.overlap * q.Hq * q.C, Mv = (double)B * (W / c.overlap);
    const bool special = h->pvq_up_kernel;     // ESCX_PVQ_UP_KERNEL=0: the GEMM engine's generic form (A/B, fallback)
    if (q.tab && special) {          // table-row add (Quant::tab): no contraction at run time, bit-identical to the kernels below
        PROF("pvq_tab_add", 0, Mv * vec * (dec ? 3 : 2) * 4,
             pvq_tab_add(codes, bstride, q.tab, q.gq, c.group_size, c.codebook_size, B, q.Hq, W, q.Cp, c.overlap, dec, out, st));
        return launch_ok("pvq_decode");
    }
    int urc = -1;
    if (special)
        PROF("pvq_up_gemm", 2.0 * Mv * vec * q.d, Mv * vec * (dec ? 2 : 1) * 4,
             urc = pvq_up(codes, bstride, q.cbraw, c.group_size, c.codebook_size, q.dt, B, q.Hq, W, q.Cp, c.overlap, q.wup, q.Kq, q.Kup, dec, out, st));
    if (urc != 0) {
        if (special && h->prof && !h->prof_recs.empty()) h->prof_recs.pop_back();
    PROF("pvq_up_gemm", 2.0 * Mv * vec * q.d, Mv * vec * (dec ? 2 : 1) * 4,
         gemm_pvq_up(codes, bstride, q.cbraw, c.group_size, c.codebook_size, q.dt, B, q.Hq, W, q.Cp, c.overlap, q.wup, q.Kq, q.Kup, dec, out, st));
    }
    return launch_ok("pvq_decode");
}

// One stream of the cross-scale quantiser: codes = search(down(enc - dec)); when `out` is given also out = dec + up(codebook[codes]) (csrvq.py:15-21).
// Default: ONE launch (fused_pvq.h).  ESCX_PVQ_FUSED=0, or a geometry the fused kernel is not instantiated for: the three-launch form
// (split-K down-projection GEMM -> pvq_search -> pvq_up), whose arithmetic the fused kernel reproduces bit for bit (profiles/r5_pvq_ab.txt).
static int run_pvq_quantize(escx_handle_s* h, const Quant& q, const float* enc, const float* dec, int B, int W, long long* codes, long long bstride,
                            float* loss, float* out, hipStream_t st) {
    const escx_config& c = h->cfg;
    if (h->pvq_fused) {
        const int Tq = W / c.overlap, M = B * Tq;
        const double vec = (double)c.overlap * q.Hq * q.C;
        int frc = -1;
        PROF(out ? "pvq_fused" : "pvq_fused_codes", 2.0 * M * vec * q.d * (out ? 2 : 1) + 2.0 * M * c.group_size * c.codebook_size * q.d,
             (double)M * vec * ((dec ? 2 : 1) + (out ? (dec ? 2 : 1) : 0)) * 4,
             frc = pvq_fused(enc, dec, B, q.Hq, W, q.Cp, c.overlap, q.wdf, q.Nz, q.Kq, pvq_down_splits(M, q.Kq, q.Cp), pvq_down_bk(q.Cp), q.cbn, q.c2, q.cbraw,
                             c.group_size, c.codebook_size, q.d, q.dt, q.wup, q.tab, q.gq, out, codes, bstride, loss, 1.0f / ((float)Tq * q.d * c.group_size), c.l2norm, st));
        if (frc == 0) return launch_ok("pvq_quantize");
        if (h->prof && !h->prof_recs.empty()) h->prof_recs.pop_back();
    }
    int rc;
    if ((rc = run_pvq_encode(h, q, enc, dec, B, W, codes, bstride, loss, st))) return rc;
    if (out && (rc = run_pvq_decode(h, q, codes, bstride, dec, B, W, out, st))) return rc;
    return 0;
}

static int run_deembed(escx_handle_s* h, const float* tok, int B, int W, float* rspec, hipStream_t st) {   // scale.py:73-81
    const escx_config& c = h->cfg;
    const int H0 = c.in_freq / c.patch_f;
    if (!h->deembed_two_stage && c.in_dim * h->Q <= 16) {
        const double tk = (double)B * H0 * W;
        int hrc = -1;
        if (h->deembed_halo)
            PROF("deembed_composed7x7", 2.0 * tk * 49 * h->C0 * c.in_dim * h->Q, (tk * h->C0 + tk * c.in_dim * h->Q) * 4,
                 hrc = deembed7_fused(tok, B, H0, W, h->C0p, h->dch_w, h->dcc_b, rspec, c.patch_f, c.patch_t, c.in_dim, h->Fp, st, h->dch_x2));
        if (hrc != 0)
        PROF("deembed_composed7x7", 2.0 * tk * 49 * h->C0 * c.in_dim * h->Q, (tk * h->C0 + tk * c.in_dim * h->Q) * 4,
             gemm_deembed_composed(tok, B, H0, W, h->C0p, h->dcc_w, rspec, h->dcc_b, c.patch_f, c.patch_t, c.in_dim, h->Fp, st));
        int brc = 0;
        PROF("deembed_border", 0, 0,
             brc = deembed_border(tok, h->dcv_w, h->dcv_b, rspec, B, H0, W, h->C0, h->C0p, c.patch_f, c.patch_t, c.in_dim, h->Fp, st));
        if (brc == 0) return launch_ok("patch_deembed");
    }
    const double toks = (double)B * H0 * W, pix = toks * h->Q;
    PROF("deembed_conv5x5", 2.0 * toks * 25 * h->C0 * h->C0 * h->Q, (toks * h->C0 + pix * h->C0) * 4,
         gemm_conv_deembed1(tok, B, H0, W, h->C0p, h->dc1_w, h->Q * h->C0p, h->deemb, h->dc1_b, c.patch_f, c.patch_t, st));
    PROF("deembed_conv3x3", 2.0 * pix * 9 * h->C0 * c.in_dim, (pix * h->C0 + pix * c.in_dim) * 4,
         gemm_conv_spec(h->deemb, B, c.patch_t * W, c.patch_f * H0, h->C0p, h->dc2_w, rspec, h->dc2_b, h->Fp, c.in_dim, st));
    return launch_ok("patch_deembed");
}

static int run_istft(escx_handle_s* h, const float* rspec, int B, int T2, float* wave, hipStream_t st) {  // base.py:39-47
    const escx_config& c = h->cfg;
    PROF("istft_idft_gemm", 2.0 * B * T2 * c.win_length * 2 * h->F, (double)B * T2 * (2 * h->F + c.win_length) * 4,
         gemm_store(rspec, c.in_dim * h->Fp, B * T2, h->idft_w, h->winP, c.in_dim * h->Fp, h->frames, h->winP, nullptr, st));
    PROF("istft_overlap_add", 0, (double)B * T2 * c.win_length * 4 + (double)B * c.hop_length * (T2 - 1) * 4,
         istft_ola(h->frames, h->win2, wave, B, T2, h->winP, c.win_length, c.hop_length, h->left, h->n_fft / 2, c.hop_length * (T2 - 1), st));
    return launch_ok("istft");
}

// frame-major padded spectrum [rows][in_dim*Fp] <-> reference-ordered [rows][in_dim][F]
static void spec_unpad(escx_handle_s* h, const float* src, float* dst, long long rows, hipStream_t st) {
    unpad_rows(src, dst, rows * h->cfg.in_dim, h->F, h->Fp, st);
}
static void spec_pad(escx_handle_s* h, const float* src, float* dst, long long rows, hipStream_t st) {
    pad_rows(src, dst, rows * h->cfg.in_dim, h->F, h->Fp, st);
}

extern "C" int escx_num_frames(escx_handle h, int n_samples) { return h ? 1 + n_samples / h->cfg.hop_length : 0; }
extern "C" int escx_output_samples(escx_handle h, int feat_w) { return h ? h->cfg.hop_length * (h->cfg.patch_t * feat_w - 1) : 0; }

// csrvq.py:131-158
static int run_csvq_encode(escx_handle_s* h, const Shapes& s, int S, long long* codes, hipStream_t st) {
    const escx_config& c = h->cfg;
    const int n = h->n, G = c.group_size;
    const long long bstride = (long long)S * G * s.Tq, sstride = (long long)G * s.Tq;
    int rc, H = s.encH[n - 1], Hn;
    float* dec = h->decA; float* other = h->decB;
    if ((rc = run_pvq_quantize(h, h->quants[0], h->enc_hs[n - 1], nullptr, s.B, s.W, codes, bstride, nullptr, S == 1 ? nullptr : dec, st))) return rc;
    if (S == 1) return 0;
    for (int i = 0; i < S - 1; ++i) {
        const Quant& q = h->quants[i + 1];
        const bool last = (i + 2 == S);                                      // csrvq.py:151: the last requested stream only emits codes
        if ((rc = run_pvq_quantize(h, q, h->enc_hs[n - 1 - i], dec, s.B, s.W, codes + (i + 1) * sstride, bstride, nullptr, last ? nullptr : dec, st))) return rc;
        if (last) break;
        if ((rc = run_layer(h, h->layers[n + i], dec, other, s.B, H, s.W, &Hn, st))) return rc;
        std::swap(dec, other); H = Hn;
    }
    return 0;
}

extern "C" int escx_encode(escx_handle h, const float* wave, int B, int L, int S, int64_t* codes, int* fh, int* fw, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!wave || !codes) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    if (S < 1 || S > h->cfg.max_streams) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "num_streams=%d outside [1, %d]", S, h->cfg.max_streams);
    if (L <= h->n_fft / 2) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_samples=%d too short for reflect padding of %d", L, h->n_fft / 2);
    Shapes s; if ((rc = ensure_ws(h, B, frames_of(h, L), &s))) return rc;
    const long long cstride = (long long)S * h->cfg.group_size * s.Tq;
    if ((rc = ensure_pvq_tables(h, (hipStream_t)stream))) return rc;
    rc = run_halves(h, B, s.T, (hipStream_t)stream, [&](int b0, int nb, hipStream_t st) -> int {
        Shapes sp = s; sp.B = nb; int r;
        if ((r = run_stft(h, wave + (size_t)b0 * L, nb, L, sp.T, h->spec, st))) return r;
        if ((r = run_encoder(h, sp, st))) return r;
        return run_csvq_encode(h, sp, S, (long long*)codes + b0 * cstride, st);
    });
    if (rc) return rc;
    if (fh) *fh = s.encH[h->n - 1];
    if (fw) *fw = s.W;
    return ESCX_OK;
}

// csrvq.py:160-183 + codecs.py:83-94
static int run_csvq_decode(escx_handle_s* h, const long long* codes, int B, int S, int Hb, int W, float* rspec, hipStream_t st) {
    const escx_config& c = h->cfg;
    const int n = h->n, G = c.group_size, Tq = W / c.overlap;
    const long long bstride = (long long)S * G * Tq, sstride = (long long)G * Tq;
    int rc, H = Hb, Hn;
    float* dec = h->decA; float* other = h->decB;
    if ((rc = run_pvq_decode(h, h->quants[0], codes, bstride, nullptr, B, W, dec, st))) return rc;
    for (int i = 0; i + 1 < n; ++i) {
        if (i < S - 1 && (rc = run_pvq_decode(h, h->quants[i + 1], codes + (i + 1) * sstride, bstride, dec, B, W, dec, st))) return rc;
        if ((rc = run_layer(h, h->layers[n + i], dec, other, B, H, W, &Hn, st))) return rc;
        std::swap(dec, other); H = Hn;
    }
    if ((rc = run_layer(h, h->layers[2 * n - 1], dec, other, B, H, W, &Hn, st))) return rc;
    return run_deembed(h, other, B, W, rspec, st);
}

extern "C" int escx_decode(escx_handle h, const int64_t* codes, int B, int S, int fh, int fw, float* wave_out, float* recon_feat,
                           void* stream) {
    int rc = check_infer_ready(h); if (rc) return rc;
    if (!codes || !wave_out) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    const escx_config& c = h->cfg;
    if (S < 1 || S > c.max_streams) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "codes.size(1)=%d outside [1, %d]", S, c.max_streams);
    if (fw < 1 || fw % c.overlap) ESCX_FAIL(ESCX_ERR_ASSERT, "Time dimension must be multiple of overlap");
    Shapes s; if ((rc = ensure_ws(h, B, frames_for_width(h, fw), &s))) return rc;
    if (s.W != fw || s.encH[h->n - 1] != fh) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "feat_shape (%d,%d) does not match the model (%d,%d)", fh, fw, s.encH[h->n - 1], s.W);
    const int T2 = c.patch_t * fw, out_len = c.hop_length * (T2 - 1);
    const long long cstride = (long long)S * c.group_size * (fw / c.overlap);
    if ((rc = ensure_pvq_tables(h, (hipStream_t)stream))) return rc;
    rc = run_halves(h, B, s.T, (hipStream_t)stream, [&](int b0, int nb, hipStream_t st) -> int {
        int r;
        if ((r = run_csvq_decode(h, (const long long*)codes + b0 * cstride, nb, S, fh, fw, h->rspec, st))) return r;
        if ((r = run_istft(h, h->rspec, nb, T2, wave_out + (size_t)b0 * out_len, st))) return r;
        if (recon_feat) spec_unpad(h, h->rspec, recon_feat + (size_t)b0 * T2 * c.in_dim * h->F, (long long)nb * T2, st);
        return launch_ok("decode");
    });
    return rc;
}

// codecs.py:30-66 in eval mode.  Exactly one of `wave` (B, L) and `feat` (B, T, in_dim, F: the reference's x_feat (B,F,T,2)
// permuted to frame-major) is given; with `feat` the STFT is skipped (codecs.py:33-34).
static int forward_impl(escx_handle h, const float* wave, const float* feat, int B, int L, int T, int S, int64_t* codes, float* wave_out,
                        float* raw_feat, float* recon_feat, float* cm_loss, void* stream) {
    int rc = check_infer_ready(h); if (rc) return rc;
    if ((!wave && !feat) || !codes || !wave_out) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    const escx_config& c = h->cfg;
    if (S < 1 || S > c.max_streams) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "num_streams=%d outside [1, %d]", S, c.max_streams);
    if (wave && L <= h->n_fft / 2) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_samples=%d too short for reflect padding of %d", L, h->n_fft / 2);
    if (!wave && T < c.patch_t) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_frames=%d shorter than one patch", T);
    if (wave) T = frames_of(h, L);
    Shapes s; if ((rc = ensure_ws(h, B, T, &s))) return rc;
    if (s.W % c.overlap) ESCX_FAIL(ESCX_ERR_ASSERT, "Time dimension must be multiple of overlap");
    const int n = h->n, G = c.group_size;
    const long long bstride = (long long)S * G * s.Tq, sstride = (long long)G * s.Tq;
    const int T2 = c.patch_t * s.W, out_len = c.hop_length * (T2 - 1);
    if ((rc = ensure_pvq_tables(h, (hipStream_t)stream))) return rc;
    return run_halves(h, B, s.T, (hipStream_t)stream, [&](int b0, int nb, hipStream_t st) -> int {
        Shapes sp = s; sp.B = nb; int r;
        long long* cd = (long long*)codes + b0 * bstride;
        if (wave) { if ((r = run_stft(h, wave + (size_t)b0 * L, nb, L, sp.T, h->spec, st))) return r; }
        else spec_pad(h, feat + (size_t)b0 * sp.T * c.in_dim * h->F, h->spec, (long long)nb * sp.T, st);
        if (raw_feat) spec_unpad(h, h->spec, raw_feat + (size_t)b0 * sp.T * c.in_dim * h->F, (long long)nb * sp.T, st);
        if ((r = run_encoder(h, sp, st))) return r;
        // per-vector commitment terms of stream slot i go to loss_terms[i][G][nb*Tq]; reduced per clip at the end (no atomics)
        const size_t lslot = (size_t)G * nb * sp.Tq;
        float* loss = cm_loss ? h->loss_terms : nullptr;
        int n_slots = 1;
        // csrvq.py:97-129 in eval mode: stream 0, then (stream i+1, block i) pairs; untransmitted streams pass through
        int H = sp.encH[n - 1], Hn;
        float* dec = h->decA; float* other = h->decB;
        if ((r = run_pvq_quantize(h, h->quants[0], h->enc_hs[n - 1], nullptr, nb, sp.W, cd, bstride, loss, dec, st))) return r;
        for (int i = 0; i + 1 < n; ++i) {
            if (i < S - 1) {
                const Quant& q = h->quants[i + 1];
                if ((r = run_pvq_quantize(h, q, h->enc_hs[n - 1 - i], dec, nb, sp.W, cd + (i + 1) * sstride, bstride, loss ? loss + (size_t)(i + 1) * lslot : nullptr, dec, st))) return r;
                n_slots = i + 2;
            }
            if ((r = run_layer(h, h->layers[n + i], dec, other, nb, H, sp.W, &Hn, st))) return r;
            std::swap(dec, other); H = Hn;
        }
        if ((r = run_layer(h, h->layers[2 * n - 1], dec, other, nb, H, sp.W, &Hn, st))) return r;
        if ((r = run_deembed(h, other, nb, sp.W, h->rspec, st))) return r;
        if ((r = run_istft(h, h->rspec, nb, T2, wave_out + (size_t)b0 * out_len, st))) return r;
        if (recon_feat) spec_unpad(h, h->rspec, recon_feat + (size_t)b0 * T2 * c.in_dim * h->F, (long long)nb * T2, st);
        if (cm_loss) loss_reduce(loss, n_slots, G, nb * sp.Tq, sp.Tq, cm_loss + b0, st);
        return launch_ok("forward");
    });
}

extern "C" int escx_forward(escx_handle h, const float* wave, int B, int L, int S, int64_t* codes, float* wave_out, float* raw_feat,
                            float* recon_feat, float* cm_loss, void* stream) {
    if (!wave) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    return forward_impl(h, wave, nullptr, B, L, 0, S, codes, wave_out, raw_feat, recon_feat, cm_loss, stream);
}

extern "C" int escx_forward_feat(escx_handle h, const float* feat, int B, int T, int S, int64_t* codes, float* wave_out, float* recon_feat,
                                 float* cm_loss, void* stream) {
    if (!feat) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    return forward_impl(h, nullptr, feat, B, 0, T, S, codes, wave_out, nullptr, recon_feat, cm_loss, stream);
}

// ------------------------------------------------------------------------------------------------
// stage-level entry points (reference layouts in and out)
// ------------------------------------------------------------------------------------------------
extern "C" int escx_spec_transform(escx_handle h, const float* wave, int B, int L, float* spec, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!wave || !spec || B < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    if (L <= h->n_fft / 2) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_samples=%d too short for reflect padding of %d", L, h->n_fft / 2);
    const int T = frames_of(h, L);
    TmpBuf t; if (t.alloc((size_t)B * T * h->cfg.in_dim * h->Fp)) ESCX_FAIL(ESCX_ERR_HIP, "hipMalloc failed");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_stft(h, wave, B, L, T, t.p, st))) return rc;
    spec_unpad(h, t.p, spec, (long long)B * T, st);
    return launch_ok("spec_transform");
}

extern "C" int escx_audio_reconstruct(escx_handle h, const float* spec, int B, int T, float* wave, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!spec || !wave || B < 1 || T < 2) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument (need at least 2 frames)");
    TmpBuf ts, tf;
    if (ts.alloc((size_t)B * T * h->cfg.in_dim * h->Fp) || tf.alloc((size_t)B * T * h->winP)) ESCX_FAIL(ESCX_ERR_HIP, "hipMalloc failed");
    hipStream_t st = (hipStream_t)stream;
    const escx_config& c = h->cfg;
    spec_pad(h, spec, ts.p, (long long)B * T, st);
    gemm_store(ts.p, c.in_dim * h->Fp, B * T, h->idft_w, h->winP, c.in_dim * h->Fp, tf.p, h->winP, nullptr, st);
    istft_ola(tf.p, h->win2, wave, B, T, h->winP, c.win_length, c.hop_length, h->left, h->n_fft / 2, c.hop_length * (T - 1), st);
    return launch_ok("audio_reconstruct");
}

extern "C" int escx_patch_embed(escx_handle h, const float* spec, int B, int T, float* tokens, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    const escx_config& c = h->cfg;
    const int W = T / c.patch_t, H0 = c.in_freq / c.patch_f;
    if (!spec || !tokens || B < 1 || W < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    TmpBuf ts, tt;
    if (ts.alloc((size_t)B * T * c.in_dim * h->Fp) || tt.alloc((size_t)B * H0 * W * h->C0p)) ESCX_FAIL(ESCX_ERR_HIP, "hipMalloc failed");
    hipStream_t st = (hipStream_t)stream;
    spec_pad(h, spec, ts.p, (long long)B * T, st);
    if ((rc = run_patch_embed(h, ts.p, B, T, W, tt.p, st))) return rc;
    unpad_rows(tt.p, tokens, (long long)B * H0 * W, h->C0, h->C0p, st);
    return launch_ok("patch_embed");
}

// stage-level calls run the whole batch on set 0, so ask for a workspace whose HALF holds B clips
static int stage_ws_for_w(escx_handle_s* h, int B, int W, Shapes* s) {
    int rc = ensure_ws(h, h->parts * B, frames_for_width(h, W), s, B);
    if (!rc) s->B = B;
    return rc;
}

extern "C" int escx_transformer_layer(escx_handle h, int layer_id, const float* x, int B, int H, int W, float* y, int* H_out, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (layer_id < 0 || layer_id >= 2 * h->n) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "layer_id out of range");
    const Layer& L = h->layers[layer_id];
    Shapes s; if ((rc = stage_ws_for_w(h, B, W, &s))) return rc;
    if ((size_t)H > (size_t)2 * s.H0) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "H too large for this model");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_pvq_tables(h, st))) return rc;
    int Hn;
    // stage buffers are sized for the largest map of the model; H*Cp never exceeds that for valid (layer, H) pairs
    const size_t need = (size_t)B * H * W * L.Cp, cap = std::max<size_t>((size_t)B * s.H0 * s.W * h->C0p, 1);
    size_t big = 0; for (int i = 0; i < h->n; ++i) big = std::max(big, (size_t)B * s.encH[i] * s.W * rup(h->cfg.h_dims[i], 16));
    if (need > std::max(cap, big)) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "(H=%d, W=%d) is larger than any map of this layer", H, W);
    pad_rows(x, h->stageA, (long long)B * H * W, L.C, L.Cp, st);
    if ((rc = run_layer(h, L, h->stageA, h->stageB, B, H, W, &Hn, st))) return rc;
    unpad_rows(h->stageB, y, (long long)B * Hn * W, L.scale ? L.Cout : L.C, L.scale ? L.CoutP : L.Cp, st);
    if (H_out) *H_out = Hn;
    return launch_ok("transformer_layer");
}

extern "C" int escx_pvq_encode(escx_handle h, int sid, const float* enc, const float* dec, int B, int W, int64_t* codes, int64_t bstride,
                               void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (sid < 0 || sid >= h->cfg.max_streams) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "stream_id out of range");
    if (W % h->cfg.overlap) ESCX_FAIL(ESCX_ERR_ASSERT, "Time dimension must be multiple of overlap");
    const Quant& q = h->quants[sid];
    Shapes s; if ((rc = stage_ws_for_w(h, B, W, &s))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * q.Hq * W;
    if ((rc = ensure_pvq_tables(h, st))) return rc;
    pad_rows(enc, h->stageA, rows, q.C, q.Cp, st);
    if (dec) pad_rows(dec, h->stageB, rows, q.C, q.Cp, st);
    return run_pvq_quantize(h, q, h->stageA, dec ? h->stageB : nullptr, B, W, (long long*)codes, bstride, nullptr, nullptr, st);
}

extern "C" int escx_pvq_decode(escx_handle h, int sid, const int64_t* codes, int64_t bstride, const float* dec, int B, int W, float* out,
                               void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (sid < 0 || sid >= h->cfg.max_streams) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "stream_id out of range");
    if (W % h->cfg.overlap) ESCX_FAIL(ESCX_ERR_ASSERT, "Time dimension must be multiple of overlap");
    const Quant& q = h->quants[sid];
    Shapes s; if ((rc = stage_ws_for_w(h, B, W, &s))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * q.Hq * W;
    if ((rc = ensure_pvq_tables(h, st))) return rc;
    if (dec) pad_rows(dec, h->stageB, rows, q.C, q.Cp, st);
    if ((rc = run_pvq_decode(h, q, (const long long*)codes, bstride, dec ? h->stageB : nullptr, B, W, h->stageA, st))) return rc;
    unpad_rows(h->stageA, out, rows, q.C, q.Cp, st);
    return launch_ok("pvq_decode");
}

extern "C" int escx_patch_deembed(escx_handle h, const float* tokens, int B, int W, float* spec, void* stream) {
    int rc = check_infer_ready(h); if (rc) return rc;
    Shapes s; if ((rc = stage_ws_for_w(h, B, W, &s))) return rc;
    hipStream_t st = (hipStream_t)stream;
    pad_rows(tokens, h->stageA, (long long)B * s.H0 * W, h->C0, h->C0p, st);
    if ((rc = run_deembed(h, h->stageA, B, W, h->rspec, st))) return rc;
    spec_unpad(h, h->rspec, spec, (long long)B * h->cfg.patch_t * W, st);
    return launch_ok("patch_deembed");
}

extern "C" int escx_debug_mlp_trace(unsigned long long* dev_buf) { mlp_set_trace(dev_buf); return 0; }
extern "C" int escx_test_fastdiv(int n, int d) { return test_fastdiv(n, d); }
extern "C" int escx_test_math(const float* x, float* y, int64_t n, int which, void* stream) {
    test_math(x, y, n, which, (hipStream_t)stream);
    return launch_ok("test_math");
}
extern "C" int escx_test_copy_rows(const float* src, float* dst, int64_t rows, int row_floats, void* stream) {
    if (!src || !dst || rows < 1 || row_floats < 4 || row_floats % 4) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    test_copy_rows(src, dst, rows, row_floats, (hipStream_t)stream);
    return launch_ok("test_copy_rows");
}
extern "C" int escx_codes_pack10(const int64_t* codes, uint8_t* out, int64_t n, void* stream) {
    if (!codes || !out || n < 0) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    codes_pack10((const long long*)codes, out, n, (hipStream_t)stream);
    return launch_ok("codes_pack10");
}
extern "C" int escx_codes_unpack10(const uint8_t* in, int64_t* codes, int64_t n, void* stream) {
    if (!codes || !in || n < 0) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    codes_unpack10(in, (long long*)codes, n, (hipStream_t)stream);
    return launch_ok("codes_unpack10");
}
extern "C" int escx_codes_narrow(const int64_t* codes, int16_t* out, int64_t n, void* stream) {
    codes_narrow((const long long*)codes, (short*)out, n, (hipStream_t)stream);
    return launch_ok("codes_narrow");
}
extern "C" int escx_codes_widen(const int16_t* in, int64_t* codes, int64_t n, void* stream) {
    codes_widen((const short*)in, (long long*)codes, n, (hipStream_t)stream);
    return launch_ok("codes_widen");
}

