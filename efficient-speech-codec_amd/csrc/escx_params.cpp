// libescx host side, part 1 of 3 (round 6 split of escx_api.cpp: VERDICT r5 item 10): handle lifetime, geometry, parameter upload and PACKING into the padded / fragment
// layouts the kernels read (escx_create ... escx_finalize_params), index maps, workspace reservation.  The launch sequences live in escx_api.cpp, the profiler in escx_profile.cpp.
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

namespace escx {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
}
}  // namespace escx

extern "C" const char* escx_last_error(void) { return g_err.c_str(); }
extern "C" const char* escx_version(void) { return "escx 0.3 (gfx950, fp32 accumulate; escx_set_precision: fp32 MFMA | three bf16 terms | two fp16 terms)"; }

// ------------------------------------------------------------------------------------------------
// configuration -> geometry
// ------------------------------------------------------------------------------------------------
static int roundup4(int x) { return rup(x, 4); }

static void add_block_keys(std::vector<std::string>& keys, const std::string& p) {
    for (const char* k : {"norm1.weight", "norm1.bias", "attn.relative_position_bias_table", "attn.qkv.weight", "attn.qkv.bias",
                          "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.linear_1.weight",
                          "mlp.linear_1.bias", "mlp.linear_2.weight", "mlp.linear_2.bias"})
        keys.push_back(p + k);
}

static int build_geometry(escx_handle_s* h) {
    const escx_config& c = h->cfg;
    const int n = c.n_scales;
    if (n < 2 || n > ESCX_MAX_SCALES) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_scales=%d out of range", n);
    // attention.py:93-127, 246-256 are generic in window_size.  4 (every shipped yaml) runs the fused MFMA kernels; any other size in [2, 16] runs the unfused launch
    // sequence with window_attention_any_kernel (inference only: the training step exists for 4 x 4 windows).
    if (c.window_size < 2 || c.window_size > 16) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "window_size=%d outside [2, 16]", c.window_size);
    h->ws = c.window_size;
    if (c.max_streams != n) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "max_streams (%d) must equal len(h_dims) (%d): one decoder block per "
                                      "residual stream (csrvq.py:108-122)", c.max_streams, n);
    if (c.in_freq % c.patch_f != 0) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "in_freq must be divisible by patch_size[0]");
    if (c.overlap < 1 || c.group_size < 1 || c.group_size > 8) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad overlap/group_size");
    h->n = n;
    h->F = c.in_freq; h->Fp = rup(c.in_freq, 16);
    h->n_fft = (c.in_freq - 1) * 2;                               // base.py:22
    if (c.win_length > h->n_fft || c.win_length < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "win_length must be in [1, n_fft]");
    h->left = (h->n_fft - c.win_length) / 2;                      // torch.stft centres the window inside n_fft
    h->winP = rup(c.win_length, 16);
    h->C0 = c.h_dims[0]; h->C0p = rup(h->C0, 16);
    h->Kpe = rup(c.in_dim * c.patch_f * c.patch_t, 16);
    h->Q = c.patch_f * c.patch_t;

    auto make_layer = [&](const std::string& prefix, int C, int nH, int scale, int Cout) -> int {
        Layer L; L.prefix = prefix; L.C = C; L.Cp = rup(C, 16); L.nH = nH;
        if (nH < 1 || C % nH != 0) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "%s: dim %d not divisible by heads %d", prefix.c_str(), C, nH);
        L.hd = C / nH; L.hdp = roundup4(L.hd);
        if (L.hdp > 64) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "head_dim %d > 64", L.hd);
        L.Nqkv = rup(3 * nH * L.hdp, 16); L.Ko = rup(nH * L.hdp, 16);
        if (L.hd <= 8) { L.attn_mode = 1; L.n_groups = (nH + 1) / 2; }
        else if (L.hd <= 16) { L.attn_mode = 0; L.n_groups = nH; }
        else if (L.hd <= 32) { L.attn_mode = 2; L.n_groups = nH; }
        if (c.window_size != 4) { L.attn_mode = -1; L.n_groups = 0; }        // no fused attention: 16-token windows are its tile
        L.hidden = (int)(C * c.mlp_ratio); L.hiddenP = rup(L.hidden, 16);
        L.scale = scale; L.Cout = Cout; L.CoutP = rup(Cout, 16);
        L.blocks.resize(c.swin_depth);
        h->layers.push_back(L);
        return 0;
    };
    // encoder: pre_nn + blocks (base.py:124-141); decoder: blocks + post_nn with reversed dims/heads (codecs.py:24-28)
    int rc;
    if ((rc = make_layer("encoder.pre_nn.", c.h_dims[0], c.swin_heads[0], 0, c.h_dims[0]))) return rc;
    for (int i = 0; i + 1 < n; ++i)
        if ((rc = make_layer("encoder.blocks." + std::to_string(i) + ".", c.h_dims[i], c.swin_heads[i], 1, c.h_dims[i + 1]))) return rc;
    for (int j = 0; j + 1 < n; ++j)
        if ((rc = make_layer("decoder.blocks." + std::to_string(j) + ".", c.h_dims[n - 1 - j], c.swin_heads[n - 2 - j], 2,
                             c.h_dims[n - 2 - j]))) return rc;
    if ((rc = make_layer("decoder.post_nn.", c.h_dims[0], c.swin_heads[0], 0, c.h_dims[0]))) return rc;

    const int H0 = c.in_freq / c.patch_f;
    for (int s = 0; s < c.max_streams; ++s) {                     // base.py:49-69
        Quant q; q.prefix = "quantizers." + std::to_string(s) + ".";
        q.C = c.h_dims[n - 1 - std::max(s - 1, 0)]; q.Cp = rup(q.C, 16);
        q.Hq = (s == 0) ? H0 >> (c.max_streams - 1) : H0 >> (c.max_streams - s);
        if (q.Hq < 1) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "quantizer %d has in_freq 0", s);
        q.d = c.codebook_dims[s]; q.dt = roundup4(q.d);
        if (q.d < 1 || q.dt > 64) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "codebook_dim %d unsupported", q.d);
        q.Nz = rup(c.group_size * q.dt, 16); q.Kup = q.Nz;
        q.Kq = c.overlap * q.Hq * q.Cp;
        h->quants.push_back(q);
    }

    // required state_dict keys
    auto& K = h->required;
    for (int s = 0; s < c.max_streams; ++s)
        for (int g = 0; g < c.group_size; ++g) {
            const std::string p = "quantizers." + std::to_string(s) + ".";
            K.push_back(p + "vqs." + std::to_string(g) + ".embedding.weight");
            K.push_back(p + "down_projs." + std::to_string(g) + ".weight");
            K.push_back(p + "up_projs." + std::to_string(g) + ".weight");
        }
    for (const char* k : {"proj.weight", "proj.bias", "norm.weight", "norm.bias"}) K.push_back(std::string("encoder.patch_embed.") + k);
    for (const Layer& L : h->layers) {
        for (int j = 0; j < c.swin_depth; ++j) add_block_keys(K, L.prefix + "swint_blocks." + std::to_string(j) + ".");
        if (L.scale) {
            K.push_back(L.prefix + "subsample.norm.weight"); K.push_back(L.prefix + "subsample.norm.bias");
            K.push_back(L.prefix + (L.scale == 1 ? "subsample.down.weight" : "subsample.up.weight"));
        }
    }
    for (const char* k : {"de_proj1.weight", "de_proj1.bias", "de_proj2.weight", "de_proj2.bias"})
        K.push_back(std::string("decoder.patch_deembed.") + k);
    return 0;
}

// The ONE place of the inference host code that reads the environment (round 6; VERDICT r5 hygiene #14): defaults of a new handle.  Product switches are documented fallbacks and
// A/B arms the tests exercise (tests/test_gpu_parity.py test_fallback_kernel_forms_against_the_default); the precision mode has a C-ABI setter (escx_set_precision), the
// environment only names its default.  Tuning switches of measured-and-rejected kernel forms go through ESCX_TUNE_ENV (null in the default build, tune_env.h).
static int env_defaults(escx_handle_s* h) {
    auto flag = [](const char* name, bool dflt) { const char* e = getenv(name); return (e && e[0]) ? e[0] == '1' : dflt; };
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; };
    h->use_fused = !flag("ESCX_NO_FUSED", false);
    h->use_fused_attn = !flag("ESCX_NO_FUSED_ATTN", false);
    h->deembed_two_stage = flag("ESCX_DEEMBED_TWO_STAGE", false);
    h->deembed_halo = !flag("ESCX_DEEMBED_GEMM", false);
    if (getenv("ESCX_STREAMS") && getenv("ESCX_STREAMS")[0]) { h->parts = std::min(std::max(num("ESCX_STREAMS", 2), 1), (int)escx_handle_s::MAX_PARTS); h->parts_forced = true; }
    h->chunk_frames = num("ESCX_CHUNK_FRAMES", h->chunk_frames);
    h->mlp_x3_max = num("ESCX_MLP_X3", h->mlp_x3_max);
    h->attn_x3_max = num("ESCX_ATTN_X3", h->attn_x3_max);
    h->rowgemm_x3 = !off("ESCX_ROWGEMM_X3");
    h->pvq_table = !off("ESCX_PVQ_TABLE");
    h->attn_gs_tokens = num("ESCX_ATTN_GS_TOKENS", h->attn_gs_tokens);
    h->mlp_split_fold = !off("ESCX_MLP_SPLIT_FOLD");
    h->pvq_fused = !off("ESCX_PVQ_FUSED");
    h->pvq_up_kernel = !off("ESCX_PVQ_UP_KERNEL");
    h->prof_serial = flag("ESCX_PROF_SERIAL", false);
    { const char* e = ESCX_TUNE_ENV("ESCX_MLP_VARIANT"); if (e && e[0]) h->mlp_variant = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_MLP_HS"); if (e && e[0]) h->mlp_hs = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_GS"); if (e && e[0]) h->attn_gs = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_NW"); if (e && e[0]) h->attn_nw = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_NO_ATTN_PACK"); h->attn_pack = !(e && e[0] == '1'); }
    // precision DEFAULT of new handles (escx_set_precision changes it per handle): ESCX_PRECISION=fp32|bf16x3|f16x2 (or 0|3|2); ESCX_X3_TERMS=3 is the round-5 spelling of bf16x3
    if (num("ESCX_X3_TERMS", 2) == 3) h->prec = 3;
    if (const char* e = getenv("ESCX_PRECISION")) {
        if (e[0]) {
            const std::string v(e);
            if (v == "fp32" || v == "0") h->prec = 0; else if (v == "bf16x3" || v == "3") h->prec = 3; else if (v == "f16x2" || v == "2") h->prec = 2;
            else ESCX_FAIL(ESCX_ERR_INVALID_ARG, "ESCX_PRECISION=%s (fp32 | bf16x3 | f16x2)", e);
        }
    }
    return 0;
}

extern "C" int escx_create(const escx_config* cfg, int device, escx_handle* out) {
    if (!cfg || !out) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null argument");
    int ndev = 0;
    ESCX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
    escx_handle_s* h = new escx_handle_s();
    h->cfg = *cfg; h->device = device;
    int erc = env_defaults(h);
    if (erc) { delete h; return erc; }
    int rc = build_geometry(h);
    if (rc) { delete h; return rc; }
    *out = h;
    return ESCX_OK;
}

extern "C" void escx_destroy(escx_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->wts.base) (void)hipFree(h->wts.base);
    for (auto& S : h->sets) if (S.ws.base) (void)hipFree(S.ws.base);
    for (int i = 0; i < escx_handle_s::MAX_PARTS; ++i) {
        if (h->sx[i]) (void)hipStreamDestroy(h->sx[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (auto& kv : h->maps) (void)hipFree(kv.second);
    if (h->coll_buf) (void)hipFree(h->coll_buf);
    for (Quant& q : h->quants) if (q.tab) (void)hipFree(q.tab);
    for (Layer& L : h->layers) { if (L.sub_x3_buf) (void)hipFree(L.sub_x3_buf); if (L.sub_x3s_buf) (void)hipFree(L.sub_x3s_buf); }
    if (h->dch_x2_buf) (void)hipFree(h->dch_x2_buf);
    for (Layer& L : h->layers) for (BlockW& bw : L.blocks) { if (bw.x3w_buf) (void)hipFree(bw.x3w_buf); if (bw.x3a_buf) (void)hipFree(bw.x3a_buf); if (bw.x3w_train) (void)hipFree(bw.x3w_train); }
    if (h->iota_codes) (void)hipFree(h->iota_codes);
    if (h->gmap) (void)hipFree(h->gmap);
    if (h->garena) (void)hipFree(h->garena);
    if (h->grad_seg) (void)hipFree(h->grad_seg);
    if (h->tape.base) (void)hipFree(h->tape.base);
    free_train_state(h);
    delete h;
}

extern "C" int escx_num_required_keys(escx_handle h) { return h ? (int)h->required.size() : 0; }
extern "C" const char* escx_required_key(escx_handle h, int i) {
    return (h && i >= 0 && i < (int)h->required.size()) ? h->required[i].c_str() : nullptr;
}

extern "C" int escx_set_param(escx_handle h, const char* key, const float* host, const int64_t* shape, int ndim) {
    if (!h || !key || !host || (ndim > 0 && !shape)) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null argument");
    Param p; size_t n = 1;
    for (int i = 0; i < ndim; ++i) { p.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    p.data.assign(host, host + n);
    h->params[key] = std::move(p);
    h->finalized = false;
    return ESCX_OK;
}

// ------------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------------
namespace {
struct Packer {
    escx_handle_s* h;
    std::vector<float> host;                    // staging image of the weight arena
    std::string missing;
    const Param* get(const std::string& key, std::initializer_list<int64_t> shape) {
        auto it = h->params.find(key);
        if (it == h->params.end()) { if (missing.empty()) missing = "missing key " + key; return nullptr; }
        const Param& p = it->second;
        if (p.shape != std::vector<int64_t>(shape)) {
            if (missing.empty()) missing = "shape mismatch for " + key;
            return nullptr;
        }
        return &p;
    }
    size_t alloc(size_t n) { size_t off = (host.size() + 63) / 64 * 64; host.resize(off + n, 0.f); return off; }
};

inline double hann(int k, int n) { return 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)k / (double)n); }

// MFMA fragment order: element (tn, kk, lane = 16*g + i, j) = W[16 tn + i][16 kk + 4 g + j]; one (tn, kk) block is
// the 1 KiB a wave fetches with a single coalesced 16-byte-per-lane load.
template <class F>
void pack_frag(float* dst, int n_tiles, int k_tiles, F at) {
    for (int tn = 0; tn < n_tiles; ++tn) for (int kk = 0; kk < k_tiles; ++kk) for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j)
            dst[((((size_t)tn * k_tiles + kk) * 64) + 16 * g + i) * 4 + j] = at(16 * tn + i, 16 * kk + 4 * g + j);
}
}  // namespace

#define GETP(var, key, ...) const Param* var = pk.get(key, {__VA_ARGS__}); if (!var) ESCX_FAIL(ESCX_ERR_STATE, "%s", pk.missing.c_str())

// Builds the host image of the weight arena from h->params.  `fix` receives (pointer slot, offset) pairs; `computed` the regions whose
// contents are NOT plain copies of parameter elements (normalised codebooks, DFT matrices, the composed de-embedding); `grads` the
// primary training layouts, i.e. the regions a backward pass produces gradients for (one-to-one with parameter elements).
static int pack_image(escx_handle_s* h, std::vector<float>& image, std::vector<std::pair<float**, size_t>>& fix,
                      std::vector<std::pair<size_t, size_t>>& computed, std::vector<std::pair<size_t, size_t>>& grads) {
    const escx_config& c = h->cfg;
    Packer pk{h};
    auto slot = [&](float** dst, size_t n) -> size_t { size_t off = pk.alloc(n); fix.push_back({dst, off}); return off; };
    auto cslot = [&](float** dst, size_t n) -> size_t { size_t off = slot(dst, n); computed.push_back({off, n}); return off; };
    auto gslot = [&](float** dst, size_t n) -> size_t { size_t off = slot(dst, n); grads.push_back({off, n}); return off; };

    // ---- transformer layers ----
    for (Layer& L : h->layers) {
        const int C = L.C, Cp = L.Cp, nH = L.nH, hd = L.hd, hdp = L.hdp;
        for (int j = 0; j < c.swin_depth; ++j) {
            const std::string p = L.prefix + "swint_blocks." + std::to_string(j) + ".";
            BlockW& bw = L.blocks[j];
            GETP(n1w, p + "norm1.weight", C); GETP(n1b, p + "norm1.bias", C);
            const int ws = c.window_size, NW2 = ws * ws;
            GETP(tab, p + "attn.relative_position_bias_table", (2 * ws - 1) * (2 * ws - 1), nH);
            GETP(qw, p + "attn.qkv.weight", 3 * C, C); GETP(qb, p + "attn.qkv.bias", 3 * C);
            GETP(pw, p + "attn.proj.weight", C, C); GETP(pb, p + "attn.proj.bias", C);
            GETP(n2w, p + "norm2.weight", C); GETP(n2b, p + "norm2.bias", C);
            GETP(w1, p + "mlp.linear_1.weight", L.hidden, C); GETP(b1, p + "mlp.linear_1.bias", L.hidden);
            GETP(w2, p + "mlp.linear_2.weight", C, L.hidden); GETP(b2, p + "mlp.linear_2.bias", C);
            size_t o;
            o = gslot(&bw.ln1_g, Cp); std::copy(n1w->data.begin(), n1w->data.end(), pk.host.begin() + o);
            o = gslot(&bw.ln1_b, Cp); std::copy(n1b->data.begin(), n1b->data.end(), pk.host.begin() + o);
            o = gslot(&bw.wqkv, (size_t)L.Nqkv * Cp);
            size_t ob = gslot(&bw.bqkv, L.Nqkv);
            for (int w = 0; w < 3; ++w) for (int hh = 0; hh < nH; ++hh) for (int d = 0; d < hd; ++d) {
                const int src = w * C + hh * hd + d, dst = w * nH * hdp + hh * hdp + d;
                std::copy(qw->data.begin() + (size_t)src * C, qw->data.begin() + (size_t)(src + 1) * C, pk.host.begin() + o + (size_t)dst * Cp);
                pk.host[ob + dst] = qb->data[src];
            }
            // relative position bias gathered per head: index = (dh + ws - 1) * (2 ws - 1) + (dw + ws - 1)  (attention.py:195-205)
            o = slot(&bw.bias_tab, (size_t)nH * NW2 * NW2);
            for (int hh = 0; hh < nH; ++hh) for (int i = 0; i < NW2; ++i) for (int jj = 0; jj < NW2; ++jj) {
                const int idx = ((i / ws) - (jj / ws) + ws - 1) * (2 * ws - 1) + ((i % ws) - (jj % ws) + ws - 1);
                pk.host[o + ((size_t)hh * NW2 + i) * NW2 + jj] = tab->data[(size_t)idx * nH + hh];
            }
            o = gslot(&bw.wproj, (size_t)Cp * L.Ko);
            for (int r = 0; r < C; ++r) for (int hh = 0; hh < nH; ++hh) for (int d = 0; d < hd; ++d)
                pk.host[o + (size_t)r * L.Ko + hh * hdp + d] = pw->data[(size_t)r * C + hh * hd + d];
            o = gslot(&bw.bproj, Cp); std::copy(pb->data.begin(), pb->data.end(), pk.host.begin() + o);
            o = gslot(&bw.ln2_g, Cp); std::copy(n2w->data.begin(), n2w->data.end(), pk.host.begin() + o);
            o = gslot(&bw.ln2_b, Cp); std::copy(n2b->data.begin(), n2b->data.end(), pk.host.begin() + o);
            o = gslot(&bw.w1, (size_t)L.hiddenP * Cp);
            for (int r = 0; r < L.hidden; ++r) std::copy(w1->data.begin() + (size_t)r * C, w1->data.begin() + (size_t)(r + 1) * C, pk.host.begin() + o + (size_t)r * Cp);
            o = gslot(&bw.b1, L.hiddenP); std::copy(b1->data.begin(), b1->data.end(), pk.host.begin() + o);
            o = gslot(&bw.w2, (size_t)Cp * L.hiddenP);
            for (int r = 0; r < C; ++r) std::copy(w2->data.begin() + (size_t)r * L.hidden, w2->data.begin() + (size_t)(r + 1) * L.hidden, pk.host.begin() + o + (size_t)r * L.hiddenP);
            o = gslot(&bw.b2, Cp); std::copy(b2->data.begin(), b2->data.end(), pk.host.begin() + o);
            {   // transposed copies for the dX GEMMs of the training step (gemm_engine computes A . W^T with W stored [N][K])
                size_t ot = slot(&bw.wqkvT, (size_t)Cp * L.Nqkv);
                for (int w = 0; w < 3; ++w) for (int hh = 0; hh < nH; ++hh) for (int d = 0; d < hd; ++d) {
                    const int src = w * C + hh * hd + d, dst = w * nH * hdp + hh * hdp + d;
                    for (int k = 0; k < C; ++k) pk.host[ot + (size_t)k * L.Nqkv + dst] = qw->data[(size_t)src * C + k];
                }
                ot = slot(&bw.wprojT, (size_t)L.Ko * Cp);
                for (int r = 0; r < C; ++r) for (int hh = 0; hh < nH; ++hh) for (int d = 0; d < hd; ++d)
                    pk.host[ot + (size_t)(hh * hdp + d) * Cp + r] = pw->data[(size_t)r * C + hh * hd + d];
                ot = slot(&bw.w1T, (size_t)Cp * L.hiddenP);
                for (int r = 0; r < L.hidden; ++r) for (int k = 0; k < C; ++k) pk.host[ot + (size_t)k * L.hiddenP + r] = w1->data[(size_t)r * C + k];
                ot = slot(&bw.w2T, (size_t)L.hiddenP * Cp);
                for (int r = 0; r < C; ++r) for (int k = 0; k < L.hidden; ++k) pk.host[ot + (size_t)k * Cp + r] = w2->data[(size_t)r * L.hidden + k];
            }
            if (L.attn_mode >= 0) {   // fused attention stream: per head group the Q, K, V and projection tiles in fragment order
                const int mode = L.attn_mode, NG = L.n_groups, KK = Cp / 16;
                const int TPG = mode == 2 ? 8 : 4, NBr = mode == 2 ? 6 : 3;
                // (head, dim) addressed by row/k-slot i of half-tile `half` of group g; -1 when padding
                auto hd_of = [&](int g, int half, int i, int* hh, int* dd) {
                    if (mode == 0) { *hh = g; *dd = i; }
                    else if (mode == 1) { *hh = 2 * g + (i >> 3); *dd = i & 7; }
                    else { *hh = g; *dd = 16 * half + i; }
                    return *hh < nH && *dd < hd;
                };
                size_t ow = slot(&bw.waf, (size_t)NG * TPG * KK * 256), obb = slot(&bw.baf, (size_t)NG * NBr * 16);
                size_t obt = slot(&bw.bias_tab_f, (size_t)(mode == 1 ? 2 * NG : NG) * 256);
                for (int hh = 0; hh < nH; ++hh) for (int i2 = 0; i2 < 16; ++i2) for (int jj = 0; jj < 16; ++jj) {
                    const int idx = ((i2 >> 2) - (jj >> 2) + 3) * 7 + ((i2 & 3) - (jj & 3) + 3);
                    pk.host[obt + ((size_t)hh * 16 + i2) * 16 + jj] = tab->data[(size_t)idx * nH + hh];
                }
                for (int g = 0; g < NG; ++g) {
                    // tile order: mode 0/1 [Q,K,V,P]; mode 2 [Q_lo,K_lo,Q_hi,K_hi,V_lo,P_lo,V_hi,P_hi]
                    struct T { int which; int half; };           // which: 0 q, 1 k, 2 v, 3 proj
                    const T order4[4] = {{0, 0}, {1, 0}, {2, 0}, {3, 0}};
                    const T order8[8] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {2, 0}, {3, 0}, {2, 1}, {3, 1}};
                    for (int ti = 0; ti < TPG; ++ti) {
                        const T tt = mode == 2 ? order8[ti] : order4[ti];
                        float* dstp = pk.host.data() + ow + ((size_t)g * TPG + ti) * KK * 256;
                        if (tt.which < 3) {
                            pack_frag(dstp, 1, KK, [&](int i2, int kx) {
                                int hh, dd; if (!hd_of(g, tt.half, i2, &hh, &dd) || kx >= C) return 0.f;
                                return qw->data[(size_t)(tt.which * C + hh * hd + dd) * C + kx]; });
                            // bias rows: mode 0/1 [q,k,v]; mode 2 [q_lo,k_lo,q_hi,k_hi,v_lo,v_hi]
                            const int brow = mode == 2 ? (tt.which == 2 ? 4 + tt.half : 2 * tt.half + tt.which) : tt.which;
                            for (int i2 = 0; i2 < 16; ++i2) {
                                int hh, dd;
                                pk.host[obb + ((size_t)g * NBr + brow) * 16 + i2] = hd_of(g, tt.half, i2, &hh, &dd) ? qb->data[tt.which * C + hh * hd + dd] : 0.f;
                            }
                        } else {    // projection: N = Cp output tiles (to), K = this tile's 16 k-slots
                            for (int to = 0; to < KK; ++to) for (int gq = 0; gq < 4; ++gq) for (int i2 = 0; i2 < 16; ++i2) for (int j = 0; j < 4; ++j) {
                                int hh, dd; const int n = 16 * to + i2;
                                const bool ok = hd_of(g, tt.half, 4 * gq + j, &hh, &dd) && n < C;
                                dstp[(((size_t)to * 64) + 16 * gq + i2) * 4 + j] = ok ? pw->data[(size_t)n * C + hh * hd + dd] : 0.f;
                            }
                        }
                    }
                }
            }
            {   // fragment-ordered copies for the fused MLP kernel
                const int hid = L.hidden;
                o = slot(&bw.w1f, (size_t)L.hiddenP * Cp);
                pack_frag(pk.host.data() + o, L.hiddenP / 16, Cp / 16, [&](int n, int k) { return (n < hid && k < C) ? w1->data[(size_t)n * C + k] : 0.f; });
                o = slot(&bw.w2f, (size_t)Cp * L.hiddenP);
                pack_frag(pk.host.data() + o, Cp / 16, L.hiddenP / 16, [&](int n, int k) { return (n < C && k < hid) ? w2->data[(size_t)n * hid + k] : 0.f; });
                // combined per-hidden-tile stream: [ht][ KK fc1 blocks (kk) | KK fc2 blocks (to) ][64][4]
                const int KK = Cp / 16, HT = L.hiddenP / 16;
                const size_t o1 = fix[fix.size() - 2].second, o2 = fix[fix.size() - 1].second;
                o = slot(&bw.wcf, (size_t)2 * L.hiddenP * Cp);
                for (int ht = 0; ht < HT; ++ht) {
                    for (int kk = 0; kk < KK; ++kk)
                        std::copy(pk.host.begin() + o1 + ((size_t)ht * KK + kk) * 256, pk.host.begin() + o1 + ((size_t)ht * KK + kk + 1) * 256,
                                  pk.host.begin() + o + ((size_t)ht * 2 * KK + kk) * 256);
                    for (int to = 0; to < KK; ++to)
                        std::copy(pk.host.begin() + o2 + ((size_t)to * HT + ht) * 256, pk.host.begin() + o2 + ((size_t)to * HT + ht + 1) * 256,
                                  pk.host.begin() + o + ((size_t)ht * 2 * KK + KK + to) * 256);
                }
            }
        }
        if (L.scale == 1) {          // PatchMerge: norm over [s][C] -> [s][Cp]; down.weight [Cout][2C] -> [CoutP][2Cp]
            GETP(nw, L.prefix + "subsample.norm.weight", 2 * C); GETP(nb, L.prefix + "subsample.norm.bias", 2 * C);
            GETP(dw, L.prefix + "subsample.down.weight", L.Cout, 2 * C);
            size_t og = gslot(&L.sub_g, 2 * Cp), ob = gslot(&L.sub_b, 2 * Cp), ow = gslot(&L.sub_w, (size_t)L.CoutP * 2 * Cp);
            for (int s = 0; s < 2; ++s) for (int cc = 0; cc < C; ++cc) {
                pk.host[og + s * Cp + cc] = nw->data[s * C + cc]; pk.host[ob + s * Cp + cc] = nb->data[s * C + cc];
                for (int r = 0; r < L.Cout; ++r) pk.host[ow + (size_t)r * 2 * Cp + s * Cp + cc] = dw->data[(size_t)r * 2 * C + s * C + cc];
            }
            {
                size_t ot = slot(&L.sub_wT, (size_t)2 * Cp * L.CoutP);          // [2Cp][CoutP]
                for (int s2 = 0; s2 < 2; ++s2) for (int cc = 0; cc < C; ++cc) for (int r = 0; r < L.Cout; ++r)
                    pk.host[ot + (size_t)(s2 * Cp + cc) * L.CoutP + r] = dw->data[(size_t)r * 2 * C + s2 * C + cc];
            }
            {
                const int Cout = L.Cout;
                size_t of = slot(&L.sub_wf, (size_t)L.CoutP * 2 * Cp);
                pack_frag(pk.host.data() + of, L.CoutP / 16, 2 * Cp / 16, [&](int n, int k) {
                    const int s2 = k / Cp, cc = k - s2 * Cp;
                    return (n < Cout && cc < C) ? dw->data[(size_t)n * 2 * C + s2 * C + cc] : 0.f; });
            }
        } else if (L.scale == 2) {   // PatchSplit: up.weight [2*Cout][C] -> [2*CoutP][Cp]
            GETP(nw, L.prefix + "subsample.norm.weight", C); GETP(nb, L.prefix + "subsample.norm.bias", C);
            GETP(uw, L.prefix + "subsample.up.weight", 2 * L.Cout, C);
            size_t og = gslot(&L.sub_g, Cp), ob = gslot(&L.sub_b, Cp), ow = gslot(&L.sub_w, (size_t)2 * L.CoutP * Cp);
            std::copy(nw->data.begin(), nw->data.end(), pk.host.begin() + og);
            std::copy(nb->data.begin(), nb->data.end(), pk.host.begin() + ob);
            for (int s = 0; s < 2; ++s) for (int r = 0; r < L.Cout; ++r)
                std::copy(uw->data.begin() + (size_t)(s * L.Cout + r) * C, uw->data.begin() + (size_t)(s * L.Cout + r + 1) * C,
                          pk.host.begin() + ow + (size_t)(s * L.CoutP + r) * Cp);
            {
                size_t ot = slot(&L.sub_wT, (size_t)Cp * 2 * L.CoutP);          // [Cp][2*CoutP]
                for (int s2 = 0; s2 < 2; ++s2) for (int r = 0; r < L.Cout; ++r) for (int k = 0; k < C; ++k)
                    pk.host[ot + (size_t)k * 2 * L.CoutP + s2 * L.CoutP + r] = uw->data[(size_t)(s2 * L.Cout + r) * C + k];
            }
            {
                const int Cout = L.Cout, CoutP = L.CoutP;
                size_t of = slot(&L.sub_wf, (size_t)2 * CoutP * Cp);
                pack_frag(pk.host.data() + of, 2 * CoutP / 16, Cp / 16, [&](int n, int k) {
                    const int s2 = n / CoutP, r = n - s2 * CoutP;
                    return (r < Cout && k < C) ? uw->data[(size_t)(s2 * Cout + r) * C + k] : 0.f; });
            }
        }
    }

    // ---- patch embed / de-embed ----
    {
        const int C0 = h->C0, C0p = h->C0p, Kin = c.in_dim * c.patch_f * c.patch_t, Q = h->Q;
        GETP(w, "encoder.patch_embed.proj.weight", C0, c.in_dim, c.patch_f, c.patch_t);
        GETP(b, "encoder.patch_embed.proj.bias", C0);
        GETP(g, "encoder.patch_embed.norm.weight", C0); GETP(be, "encoder.patch_embed.norm.bias", C0);
        size_t o = gslot(&h->pe_w, (size_t)C0p * h->Kpe);
        for (int r = 0; r < C0; ++r) std::copy(w->data.begin() + (size_t)r * Kin, w->data.begin() + (size_t)(r + 1) * Kin, pk.host.begin() + o + (size_t)r * h->Kpe);
        o = gslot(&h->pe_b, C0p); std::copy(b->data.begin(), b->data.end(), pk.host.begin() + o);
        o = gslot(&h->pe_g, C0p); std::copy(g->data.begin(), g->data.end(), pk.host.begin() + o);
        o = gslot(&h->pe_beta, C0p); std::copy(be->data.begin(), be->data.end(), pk.host.begin() + o);

        GETP(w1, "decoder.patch_deembed.de_proj1.weight", (int64_t)C0 * Q, C0, 5, 5);
        GETP(b1, "decoder.patch_deembed.de_proj1.bias", (int64_t)C0 * Q);
        GETP(w2, "decoder.patch_deembed.de_proj2.weight", c.in_dim, C0, 3, 3);
        GETP(b2, "decoder.patch_deembed.de_proj2.bias", c.in_dim);
        const size_t K1 = (size_t)25 * C0p;
        size_t ow = gslot(&h->dc1_w, (size_t)Q * C0p * K1), ob = gslot(&h->dc1_b, (size_t)Q * C0p);
        // pixel_shuffle splits the conv channel dim as (s1, s2, C): o = q*C0 + co  (scale.py:16-23,78)
        for (int q = 0; q < Q; ++q) for (int co = 0; co < C0; ++co) {
            const size_t orow = (size_t)q * C0 + co, prow = (size_t)q * C0p + co;
            pk.host[ob + prow] = b1->data[orow];
            for (int ci = 0; ci < C0; ++ci) for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw)
                pk.host[ow + prow * K1 + (size_t)(kh * 5 + kw) * C0p + ci] = w1->data[((orow * C0 + ci) * 5 + kh) * 5 + kw];
        }
        {   // conv5x5 dX as an implicit GEMM over (tap, q, co): W'[ci][(tap*Q + q)*C0p + co] = w1[q*C0 + co][ci][kh][kw], tap = kh*5 + kw
            size_t ot = slot(&h->dc1_wT, (size_t)C0p * 25 * Q * C0p);
            for (int q = 0; q < Q; ++q) for (int co = 0; co < C0; ++co) for (int ci = 0; ci < C0; ++ci) for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw)
                pk.host[ot + (size_t)ci * 25 * Q * C0p + (size_t)((kh * 5 + kw) * Q + q) * C0p + co] =
                    w1->data[((((size_t)q * C0 + co) * C0 + ci) * 5 + kh) * 5 + kw];
        }
        // conv2 runs on the TIME-major map (D0 = time, D1 = freq): tap (t0 over time = kw, t1 over freq = kh)
        const size_t K2 = (size_t)9 * C0p;
        ow = gslot(&h->dc2_w, (size_t)16 * K2); ob = gslot(&h->dc2_b, 16);
        if (c.in_dim > 4) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "in_dim > 4");
        for (int oc = 0; oc < c.in_dim; ++oc) {
            pk.host[ob + oc] = b2->data[oc];
            for (int ci = 0; ci < C0; ++ci) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw)
                pk.host[ow + (size_t)oc * K2 + (size_t)(kw * 3 + kh) * C0p + ci] = w2->data[(((size_t)oc * C0 + ci) * 3 + kh) * 3 + kw];
        }
    }

    // ---- composed de-embedding: conv3x3 o pixel_shuffle o conv5x5 has no non-linearity in between (scale.py:73-81), so it is ONE
    //      linear map from a 7x7 coarse neighbourhood (x C0) to the in_dim*pf*pt fine outputs of a coarse pixel: 11x fewer FLOPs
    //      than the two convolutions (the 270-channel expansion collapses).  Folded in fp64.  The 3x3 zero-pads the FINE map, so
    //      coarse pixels on the first/last row/column use variants that drop the out-of-range fine neighbours.
    {
        const int C0 = h->C0, C0p = h->C0p, Q = h->Q, pf = c.patch_f, pt = c.patch_t, NO = c.in_dim * Q;
        const Param* w1 = &h->params["decoder.patch_deembed.de_proj1.weight"]; const Param* b1 = &h->params["decoder.patch_deembed.de_proj1.bias"];
        const Param* w2 = &h->params["decoder.patch_deembed.de_proj2.weight"]; const Param* b2 = &h->params["decoder.patch_deembed.de_proj2.bias"];
        const size_t Kc = (size_t)49 * C0;
        std::vector<double> wc((size_t)16 * NO * Kc, 0.0), bc((size_t)16 * NO, 0.0);
        auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        for (int eh = 0; eh < 4; ++eh) for (int ew = 0; ew < 4; ++ew) {
            const int v = 4 * eh + ew;
            for (int co = 0; co < c.in_dim; ++co) for (int s1 = 0; s1 < pf; ++s1) for (int s2 = 0; s2 < pt; ++s2) {
                const int n = co * Q + s1 * pt + s2;
                double* wrow = wc.data() + ((size_t)v * NO + n) * Kc;
                bc[(size_t)v * NO + n] = b2->data[co];
                for (int a = 0; a < 3; ++a) for (int bq = 0; bq < 3; ++bq) {
                    const int dh0 = fdiv(s1 + a - 1, pf), s1n = s1 + a - 1 - dh0 * pf;
                    const int dw0 = fdiv(s2 + bq - 1, pt), s2n = s2 + bq - 1 - dw0 * pt;
                    if ((dh0 < 0 && (eh & 1)) || (dh0 > 0 && (eh & 2)) || (dw0 < 0 && (ew & 1)) || (dw0 > 0 && (ew & 2))) continue;   // fine neighbour outside
                    const int qn = s1n * pt + s2n;
                    for (int cc = 0; cc < C0; ++cc) {
                        const double w2v = w2->data[(((size_t)co * C0 + cc) * 3 + a) * 3 + bq];
                        const size_t orow = (size_t)qn * C0 + cc;
                        bc[(size_t)v * NO + n] += w2v * b1->data[orow];
                        for (int ci = 0; ci < C0; ++ci) for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw) {
                            const int dh = dh0 + kh - 2, dw = dw0 + kw - 2;
                            wrow[(size_t)((dh + 3) * 7 + (dw + 3)) * C0 + ci] += w2v * w1->data[((orow * C0 + ci) * 5 + kh) * 5 + kw];
                        }
                    }
                }
            }
        }
        size_t o = cslot(&h->dcc_w, (size_t)16 * 49 * C0p);
        for (int n = 0; n < NO; ++n) for (int tap = 0; tap < 49; ++tap) for (int ci = 0; ci < C0; ++ci)
            pk.host[o + (size_t)n * 49 * C0p + (size_t)tap * C0p + ci] = (float)wc[(size_t)n * Kc + (size_t)tap * C0 + ci];
        o = cslot(&h->dcc_b, 16);
        for (int n = 0; n < NO; ++n) pk.host[o + n] = (float)bc[n];
        // the same interior weights as MFMA fragments [tap][kk][lane][4] for the halo-tiled kernel: lane (n = l & 15, g = l >> 4), channel 16kk + 4g + r
        const int KKd = C0p / 16;
        o = cslot(&h->dch_w, (size_t)49 * KKd * 256);
        if (NO <= 16)
            for (int tap = 0; tap < 49; ++tap) for (int kk = 0; kk < KKd; ++kk) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int n = l & 15, ci = 16 * kk + 4 * (l >> 4) + r;
                pk.host[o + ((size_t)(tap * KKd + kk) * 64 + l) * 4 + r] = (n < NO && ci < C0) ? (float)wc[(size_t)n * Kc + (size_t)tap * C0 + ci] : 0.f;
            }
        o = cslot(&h->dcv_w, (size_t)16 * NO * Kc);
        for (size_t i = 0; i < (size_t)16 * NO * Kc; ++i) pk.host[o + i] = (float)wc[i];
        o = cslot(&h->dcv_b, (size_t)16 * NO);
        for (size_t i = 0; i < (size_t)16 * NO; ++i) pk.host[o + i] = (float)bc[i];
    }

    // ---- windowed DFT / inverse DFT matrices (base.py:22-47; torch.stft / torch.istft semantics) ----
    {
        const int win = c.win_length, N = h->n_fft, F = h->F, Fp = h->Fp, left = h->left;
        std::vector<double> w(win);
        auto it = h->params.find("ft.window");
        for (int k = 0; k < win; ++k) w[k] = (it != h->params.end() && (int)it->second.data.size() == win) ? (double)it->second.data[k] : hann(k, win);
        size_t o = cslot(&h->dft_w, (size_t)2 * Fp * h->winP);
        for (int f = 0; f < F; ++f) for (int k = 0; k < win; ++k) {
            const double ang = 2.0 * M_PI * (double)((long long)f * (k + left) % N) / (double)N;
            pk.host[o + (size_t)f * h->winP + k] = (float)(w[k] * std::cos(ang));
            pk.host[o + (size_t)(Fp + f) * h->winP + k] = (float)(-w[k] * std::sin(ang));
        }
        std::vector<double> wi(win);
        auto it2 = h->params.find("ift.window");
        for (int k = 0; k < win; ++k) wi[k] = (it2 != h->params.end() && (int)it2->second.data.size() == win) ? (double)it2->second.data[k] : hann(k, win);
        o = cslot(&h->idft_w, (size_t)h->winP * 2 * Fp);
        for (int j = 0; j < win; ++j) for (int f = 0; f < F; ++f) {
            const double coef = (f == 0 || (N % 2 == 0 && f == N / 2)) ? 1.0 : 2.0;   // Hermitian completion of a onesided spectrum
            const double ang = 2.0 * M_PI * (double)((long long)f * (j + left) % N) / (double)N;
            pk.host[o + (size_t)j * 2 * Fp + f] = (float)(wi[j] * coef * std::cos(ang) / N);
            pk.host[o + (size_t)j * 2 * Fp + Fp + f] = (float)(-wi[j] * coef * std::sin(ang) / N);
        }
        {   // inverse-DFT matrix transposed ([2Fp][winP]) for the waveform -> spectrum gradient
            size_t ot = cslot(&h->idft_wT, (size_t)2 * Fp * h->winP);
            const size_t oi = fix[fix.size() - 2].second;
            for (int j = 0; j < h->winP; ++j) for (int f = 0; f < 2 * Fp; ++f) pk.host[ot + (size_t)f * h->winP + j] = pk.host[oi + (size_t)j * 2 * Fp + f];
        }
        o = cslot(&h->win2, h->winP);
        for (int j = 0; j < win; ++j) { const float wf = (float)wi[j]; pk.host[o + j] = wf * wf; }
    }

    // ---- product quantisers ----
    const int G = c.group_size, Ksz = c.codebook_size;
    for (Quant& q : h->quants) {
        const int fix = q.Hq * q.C, D = c.overlap * fix;
        std::vector<int> dims(G, D / G); dims[G - 1] = D - (D / G) * (G - 1);     // quantization.py:380-386
        size_t owd = gslot(&q.wd, (size_t)q.Nz * q.Kq), owu = gslot(&q.wup, (size_t)q.Kq * q.Kup);
        size_t owdT = slot(&q.wdT, (size_t)q.Kq * q.Nz), owuT = slot(&q.wupT, (size_t)q.Kup * q.Kq);
        size_t ocn = cslot(&q.cbn, (size_t)G * Ksz * q.dt), oc2 = cslot(&q.c2, (size_t)G * Ksz), ocr = gslot(&q.cbraw, (size_t)G * Ksz * q.dt);
        size_t owf = slot(&q.wdf, (size_t)q.Nz * q.Kq), ogq = cslot(&q.gq, (size_t)q.Kq / 4);
        std::vector<int> grp_of((size_t)q.Kq, -1);                             // group of every element of the framed vector in memory order
        int start = 0;
        for (int g = 0; g < G; ++g) {
            const std::string gs = std::to_string(g);
            GETP(emb, q.prefix + "vqs." + gs + ".embedding.weight", Ksz, q.d);
            GETP(dw, q.prefix + "down_projs." + gs + ".weight", q.d, dims[g]);
            GETP(uw, q.prefix + "up_projs." + gs + ".weight", dims[g], q.d);
            for (int e = 0; e < dims[g]; ++e) {
                const int flat = start + e;                                     // (o, c, h) order: quantization.py:400-409
                const int o = flat / fix, r = flat - o * fix, cc = r / q.Hq, hh = r - cc * q.Hq;
                const size_t col = (size_t)(o * q.Hq + hh) * q.Cp + cc;         // internal (o, h, c) order
                grp_of[col] = g;
                for (int j = 0; j < q.d; ++j) {
                    pk.host[owd + (size_t)(g * q.dt + j) * q.Kq + col] = dw->data[(size_t)j * dims[g] + e];
                    pk.host[owu + col * q.Kup + g * q.dt + j] = uw->data[(size_t)e * q.d + j];
                    pk.host[owdT + col * q.Nz + g * q.dt + j] = dw->data[(size_t)j * dims[g] + e];          // [Kq][Nz]: d residual = d z_e . W_down
                    pk.host[owuT + (size_t)(g * q.dt + j) * q.Kq + col] = uw->data[(size_t)e * q.d + j];    // [Kup][Kq]: d z_up = d out . W_up
                }
            }
            for (int k = 0; k < Ksz; ++k) {
                const float* row = emb->data.data() + (size_t)k * q.d;
                float ss = 0.f;
                for (int j = 0; j < q.d; ++j) ss += row[j] * row[j];
                const float den = c.l2norm ? std::max(std::sqrt(ss), 1e-12f) : 1.0f;   // F.normalize (codebook.py:32)
                float s2 = 0.f;
                for (int j = 0; j < q.d; ++j) {
                    const float v = row[j] / den;
                    pk.host[ocn + ((size_t)g * Ksz + k) * q.dt + j] = v;
                    pk.host[ocr + ((size_t)g * Ksz + k) * q.dt + j] = row[j];
                    s2 += v * v;
                }
                pk.host[oc2 + (size_t)g * Ksz + k] = s2;
            }
            start += dims[g];
        }
        // fragment order of the down-projection for the fused kernel: (k chunk, n tile, lane = 16 * slot + i, j) = W[16 tile + i][16 chunk + 4 slot + j]
        {
            const int NT = q.Nz / 16, KC = q.Kq / 16;
            for (int ck = 0; ck < KC; ++ck) for (int tn = 0; tn < NT; ++tn) for (int sl = 0; sl < 4; ++sl) for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j)
                pk.host[owf + ((((size_t)ck * NT + tn) * 64) + 16 * sl + i) * 4 + j] = pk.host[owd + (size_t)(16 * tn + i) * q.Kq + 16 * ck + 4 * sl + j];
        }
        // group of every float4 (padding channels belong to no group); a float4 that straddles two groups rules the table form out
        q.tab_ok = true;
        for (int f4 = 0; f4 < q.Kq / 4; ++f4) {
            int g4 = -1;
            for (int e = 0; e < 4; ++e) {
                const int ge = grp_of[(size_t)4 * f4 + e];
                if (ge < 0) continue;
                if (g4 >= 0 && ge != g4) q.tab_ok = false;
                g4 = ge;
            }
            pk.host[ogq + f4] = (float)g4;
        }
    }
    h->train_x3_stale = true;
    h->pvq_tab_stale = true;

    image.swap(pk.host);
    return 0;
}

// canonical flat order of the trainable parameters: the keys escx_finalize_params requires, in that order
static void build_flat_layout(escx_handle_s* h) {
    if (!h->flat_keys.empty()) return;
    size_t off = 0;
    for (const std::string& k : h->required) {
        auto it = h->params.find(k);
        const size_t n = it == h->params.end() ? 0 : it->second.data.size();
        h->flat_keys.push_back(k); h->flat_off.push_back(off); h->flat_numel.push_back(n);
        off += n;
    }
    h->flat_total = off;
}

extern "C" int escx_finalize_params(escx_handle h) {
    if (!h) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    ESCX_HIP(hipSetDevice(h->device));
    std::vector<float> image;
    std::vector<std::pair<float**, size_t>> fix;
    std::vector<std::pair<size_t, size_t>> computed, grads;
    int rc = pack_image(h, image, fix, computed, grads);
    if (rc) return rc;
    // ---- upload (the arena is kept when its size is unchanged: re-packing after an optimiser step does not reallocate) ----
    const size_t bytes = image.size() * sizeof(float);
    if (h->wts.base && h->wts.cap != bytes) { ESCX_HIP(hipDeviceSynchronize()); ESCX_HIP(hipFree(h->wts.base)); h->wts = Arena(); }
    if (!h->wts.base) {
        ESCX_HIP(hipMalloc((void**)&h->wts.base, bytes));
        h->wts.cap = h->wts.used = bytes;
        if (h->gmap) { (void)hipFree(h->gmap); h->gmap = nullptr; }
    }
    ESCX_HIP(hipMemcpy(h->wts.base, image.data(), bytes, hipMemcpyHostToDevice));
    for (auto& f : fix) *f.first = reinterpret_cast<float*>(h->wts.base) + f.second;
    h->grad_regions = grads;
    if (h->grad_seg) { (void)hipFree(h->grad_seg); h->grad_seg = nullptr; }      // rebuilt by the next training backward
    h->finalized = true;
    build_flat_layout(h);
    for (Layer& L : h->layers)
        for (size_t j = 0; j < L.blocks.size(); ++j) {
            const std::string key = L.prefix + "swint_blocks." + std::to_string(j) + ".attn.relative_position_bias_table";
            for (size_t i = 0; i < h->flat_keys.size(); ++i) if (h->flat_keys[i] == key) L.blocks[j].tab_off = (long long)h->flat_off[i];
        }
    return ESCX_OK;
}

// Gather map of the arena (training step): the packer is run a second time on parameters whose VALUES are their own flat index + 1
// (exact in fp32 below 2^24), so every plain-copy element of the image then names the parameter element it came from; 0 = structural
// zero (padding), -1 = computed region.  One int per arena float, uploaded once per handle.
int escx::build_gather_map(escx_handle_s* h) {
    if (h->gmap) return 0;
    if (!h->finalized) ESCX_FAIL(ESCX_ERR_STATE, "parameters not finalised");
    if (h->flat_total + 1 >= (size_t)1 << 24) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "model too large for the fp32-coded gather map (%zu parameters)", h->flat_total);
    std::map<std::string, Param> saved;
    for (size_t i = 0; i < h->flat_keys.size(); ++i) {
        Param& p = h->params[h->flat_keys[i]];
        saved[h->flat_keys[i]] = p;
        for (size_t e = 0; e < p.data.size(); ++e) p.data[e] = (float)(h->flat_off[i] + e + 1);
    }
    std::vector<float> image;
    std::vector<std::pair<float**, size_t>> fix;
    std::vector<std::pair<size_t, size_t>> computed, grads;
    float* dummy_slots = nullptr; (void)dummy_slots;
    // pack_image writes pointer slots only through `fix`, which is discarded here
    int rc = pack_image(h, image, fix, computed, grads);
    for (auto& kv : saved) h->params[kv.first] = kv.second;
    if (rc) return rc;
    if (image.size() * sizeof(float) != h->wts.cap) ESCX_FAIL(ESCX_ERR_STATE, "gather map image size mismatch");
    std::vector<int> gm(image.size(), 0);
    std::vector<char> is_computed(image.size(), 0);
    for (auto& r : computed) for (size_t i = r.first; i < r.first + r.second; ++i) is_computed[i] = 1;
    for (size_t i = 0; i < image.size(); ++i) {
        if (is_computed[i]) { gm[i] = -1; continue; }
        const float v = image[i];
        const long long c = (long long)v;
        if (v < 0.f || (float)c != v || (size_t)c > h->flat_total) ESCX_FAIL(ESCX_ERR_STATE, "gather map: element %zu is not a plain copy (%g)", i, (double)v);
        gm[i] = (int)c;
    }
    ESCX_HIP(hipSetDevice(h->device));
    ESCX_HIP(hipMalloc((void**)&h->gmap, gm.size() * sizeof(int)));
    ESCX_HIP(hipMemcpy(h->gmap, gm.data(), gm.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!h->garena) ESCX_HIP(hipMalloc((void**)&h->garena, h->wts.cap));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// geometry for a batch, index maps, workspace
// ------------------------------------------------------------------------------------------------
int escx::make_shapes(escx_handle_s* h, int B, int T, Shapes* out) {
    const escx_config& c = h->cfg;
    if (B < 1 || T < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "batch and frame count must be positive");
    Shapes s; s.B = B; s.L = 0;
    s.T = T;
    s.W = s.T / c.patch_t;                      // the strided conv drops a trailing odd frame (scale.py:42)
    s.H0 = c.in_freq / c.patch_f;
    if (s.W < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "input too short");
    if (s.W % c.overlap != 0) ESCX_FAIL(ESCX_ERR_ASSERT, "Time dimension must be multiple of overlap");   // quantization.py:407
    s.Tq = s.W / c.overlap;
    int H = s.H0;
    s.encH.push_back(H);
    for (int i = 0; i + 1 < h->n; ++i) { H = (H + 1) / 2; s.encH.push_back(H); }
    // the decoder doubles H per block; residuals need matching shapes (csrvq.py:15-17) and the quantisers a fixed in_freq
    for (int st = 0; st < c.max_streams; ++st) {
        const int scale = h->n - 1 - std::max(st - 1, 0);
        if (h->quants[st].Hq != s.encH[scale])
            ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "in_freq/patch (%d) must be divisible by 2^(max_streams-1)", s.H0);
    }
    *out = s;
    return 0;
}

int escx::get_map(escx_handle_s* h, int H, int W, int shift, const int** out) {
    auto key = std::make_tuple(H, W, shift);
    auto it = h->maps.find(key);
    if (it != h->maps.end()) { *out = it->second; return 0; }
    std::vector<int> m;
    if (shift >= 10) {                          // inverse of the window map: token -> slot (LayerNorm backward of the training step)
        const int sh0 = shift - 10, ws = h->ws;
        const int Hp = rup(H, ws), Wp = rup(W, ws), nWw = Wp / ws;
        m.assign((size_t)H * W, 0);
        for (int hh = 0; hh < Hp; ++hh) for (int ww = 0; ww < Wp; ++ww) {
            const int sh = (hh + sh0) % Hp, sw = (ww + sh0) % Wp;
            const int slot = ((hh / ws) * nWw + (ww / ws)) * ws * ws + (hh % ws) * ws + (ww % ws);
            if (sh < H && sw < W) m[(size_t)sh * W + sw] = slot;
        }
    } else
    if (shift >= 0) {                           // window slots -> source token (attention.py:139-155, 246-250)
        const int ws = h->ws;
        const int Hp = rup(H, ws), Wp = rup(W, ws), nWw = Wp / ws;
        m.resize((size_t)Hp * Wp);
        for (int hh = 0; hh < Hp; ++hh) for (int ww = 0; ww < Wp; ++ww) {
            const int sh = (hh + shift) % Hp, sw = (ww + shift) % Wp;       // roll(-shift) over the PADDED map
            const int slot = ((hh / ws) * nWw + (ww / ws)) * ws * ws + (hh % ws) * ws + (ww % ws);
            m[slot] = (sh < H && sw < W) ? sh * W + sw : -1;
        }
    } else {                                    // PatchMerge rows (scale.py:104-112)
        const int H2 = (H + 1) / 2;
        m.resize((size_t)H2 * W * 2);
        for (int h2 = 0; h2 < H2; ++h2) for (int w = 0; w < W; ++w) {
            m[((size_t)h2 * W + w) * 2 + 0] = (2 * h2) * W + w;
            m[((size_t)h2 * W + w) * 2 + 1] = (2 * h2 + 1 < H) ? (2 * h2 + 1) * W + w : -1;
        }
    }
    // The cache is bounded (a caller that streams clips of many different lengths, scripts/test.py on a real data set, would otherwise grow
    // device memory without limit), but nothing is evicted HERE: a launch sequence may hold several maps at once (train.hip fetches a map and
    // its inverse back to back).  Eviction happens at entry-point boundaries only: check_ready() -> trim_maps().
    int* d = nullptr;
    ESCX_HIP(hipMalloc((void**)&d, m.size() * sizeof(int)));
    ESCX_HIP(hipMemcpy(d, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice));
    h->maps[key] = d;
    *out = d;
    return 0;
}

extern "C" int64_t escx_workspace_bytes(escx_handle h) {
    int64_t t = 0;
    if (h) for (auto& S : h->sets) t += (int64_t)S.ws.cap;
    return t;
}

void escx::use_set(escx_handle_s* h, int i) { static_cast<WsFields&>(*h) = h->sets[i]; }
// number of parts a batch of B clips is split into, and the clips one workspace set must hold
// Batches under 6 clips run as ONE part: splitting 2 - 4 clips over two streams costs more in per-launch efficiency than the overlap of the
// parts' tails returns (measured, tools/small_batch.py: B = 2 4.52 -> 4.12 ms, B = 4 4.96 -> 4.67 ms; from B = 8 up two parts win).  The
// arithmetic of a clip does not depend on how the batch is split, so the codes stay identical either way.
int escx::n_parts(escx_handle_s* h, int B) { return std::max(1, std::min((h->parts_forced || B >= 6) ? h->parts : 1, B)); }
// Clips a workspace set must hold = clips of one PASS of a part.  Round 6: a part of a large batch walks its clips in passes of at most chunk_frames / T clips, so that the
// activations a pass hands from kernel to kernel stay within reach of the 256 MB memory-side cache (18 clips x 601 frames: the C = 45 maps are 66 MB each) - at 144 clips per
// part they are 530 MB and every kernel re-reads its input from HBM, which is why 288 clips ran 4 % SLOWER per clip than 36 (profiles/r6_chunk_ab.txt).  Clips are
// independent end to end: the pass structure changes no arithmetic.
int escx::pass_clips(escx_handle_s* h, int B, int T) {
    const int k = n_parts(h, B), per = (B + k - 1) / k;
    if (h->chunk_frames <= 0 || T <= 0) return per;
    return std::max(1, std::min(per, h->chunk_frames / T));
}
static int set_clips(escx_handle_s* h, int B, int T) { return pass_clips(h, B, T); }

static bool ws_fits(escx_handle_s* h, int B, int T) {
    const int need = set_clips(h, B, T), sets = n_parts(h, B);
    for (int i = 0; i < sets; ++i) {
        const WsFields& S = h->sets[i];
        // capacity, not equality: every buffer is sized by (clips, frames) maxima and grows monotonically with both, so a shorter
        // clip or a smaller batch reuses the workspace (no hipFree/hipMalloc/synchronise per new length)
        if (!(S.ws.base && S.shp.B >= need && S.shp.T >= T)) return false;
    }
    return true;
}

// set_clips_min / sets_min: lower bounds carried over from the existing workspace when it grows (the clips a part holds are NOT monotone in
// the batch: 5 clips run as one part of 5, 8 clips as two parts of 4)
static int reserve_frames(escx_handle_s* h, int Btotal, int T, int set_clips_min = 0, int sets_min = 0) {
    ESCX_HIP(hipSetDevice(h->device));
    const int B = std::max(set_clips(h, Btotal, T), set_clips_min), sets = std::max(n_parts(h, Btotal), sets_min);
    Shapes s;
    int rc = make_shapes(h, B, T, &s);
    if (rc) return rc;
    const escx_config& c = h->cfg;
    const int n = h->n;
    // pre-build every index map the whole-path calls will need
    const int* dummy;
    for (int li = 0; li < 2 * n; ++li) {
        const Layer& Ly = h->layers[li];
        int H;
        if (li < n) H = s.encH[std::max(li - 1, 0)];
        else if (li < 2 * n - 1) H = s.encH[n - 1 - (li - n)];
        else H = s.encH[0];
        if ((rc = get_map(h, H, s.W, 0, &dummy))) return rc;
        if ((rc = get_map(h, H, s.W, h->ws / 2, &dummy))) return rc;
        if (Ly.scale == 1 && (rc = get_map(h, H, s.W, -1, &dummy))) return rc;
    }
    if (ws_fits(h, Btotal, T)) return ESCX_OK;
    // grow only: keep the largest batch and clip length seen so far, so that callers alternating between shapes do not thrash
    if (h->sets[0].ws.base && !(set_clips_min || sets_min)) {
        const int Bs = std::max(B, h->sets[0].shp.B), ns = std::max(sets, h->n_sets), Tt = std::max(T, h->sets[0].shp.T);
        if (Bs != B || ns != sets || Tt != T) {
            Shapes s2;
            if (make_shapes(h, Bs, Tt, &s2) == 0) return reserve_frames(h, Btotal, Tt, Bs, ns);
        }
    }
    if (!h->ev_fork) ESCX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 1; i < sets; ++i) if (!h->sx[i]) {
        ESCX_HIP(hipStreamCreateWithFlags(&h->sx[i], hipStreamNonBlocking));
        ESCX_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }

    // size every buffer (floats)
    size_t work = 0, xn = 0, qkv = 0, ob = 0, hid = 0, dec = 0, zp = 0;
    std::vector<size_t> ehs(n);
    for (int i = 0; i < n; ++i) ehs[i] = (size_t)B * s.encH[i] * s.W * rup(c.h_dims[i], 16);
    for (int li = 0; li < 2 * n; ++li) {
        const Layer& Ly = h->layers[li];
        int H;
        if (li < n) H = s.encH[std::max(li - 1, 0)];
        else if (li < 2 * n - 1) H = s.encH[n - 1 - (li - n)];
        else H = s.encH[0];
        const size_t tokens = (size_t)B * H * s.W, slots = (size_t)B * rup(H, h->ws) * rup(s.W, h->ws);
        work = std::max(work, tokens * Ly.Cp);
        xn = std::max({xn, slots * Ly.Cp, tokens * Ly.Cp, Ly.scale == 1 ? (size_t)B * ((H + 1) / 2) * s.W * 2 * Ly.Cp : 0});
        qkv = std::max(qkv, slots * Ly.Nqkv);
        ob = std::max(ob, slots * Ly.Ko);
        hid = std::max(hid, tokens * Ly.hiddenP);
        if (li >= n) dec = std::max({dec, tokens * Ly.Cp, (size_t)B * (Ly.scale == 2 ? 2 * H : H) * s.W * Ly.CoutP});
    }
    for (const Quant& q : h->quants) {
        const int sp = pvq_down_splits(B * s.Tq, q.Kq, q.Cp);
        zp = std::max(zp, (size_t)sp * B * s.Tq * q.Nz);
    }
    const int T2 = c.patch_t * s.W, F2 = c.patch_f * s.H0;
    const size_t spec = (size_t)B * s.T * c.in_dim * h->Fp;
    const size_t deemb = h->deembed_two_stage ? (size_t)B * T2 * F2 * h->C0p : 64;
    const size_t rspec = (size_t)B * T2 * c.in_dim * h->Fp;
    const size_t frames = (size_t)B * T2 * h->winP;
    const size_t stage = std::max({work, dec, spec, rspec});
    const size_t codes = (size_t)B * c.max_streams * c.group_size * s.Tq * 2;      // int64 as 2 floats
    size_t total = 0;
    auto add = [&](size_t nfl) { total += (nfl * sizeof(float) + 255) / 256 * 256; };
    add(spec); for (int i = 0; i < n; ++i) add(ehs[i]);
    add(work); add(xn); add(qkv); add(ob); add(hid); add(dec); add(dec); add(zp); add(deemb); add(rspec); add(frames);
    const size_t lterms = (size_t)c.max_streams * c.group_size * B * s.Tq;
    add(stage); add(stage); add(codes); add(B); add(lterms); add(WsFields::N_TICKETS);

    ESCX_HIP(hipDeviceSynchronize());
    for (int si = 0; si < escx_handle_s::MAX_PARTS; ++si) {
        WsFields& S = h->sets[si];
        if (S.ws.base) { ESCX_HIP(hipFree(S.ws.base)); }
        S = WsFields();
        if (si >= sets) continue;
        ESCX_HIP(hipMalloc((void**)&S.ws.base, total));
        ESCX_HIP(hipMemset(S.ws.base, 0, total));
        S.ws.cap = total; S.ws.used = 0;
        S.spec = S.ws.take(spec);
        S.enc_hs.assign(n, nullptr);
        for (int i = 0; i < n; ++i) S.enc_hs[i] = S.ws.take(ehs[i]);
        S.work = S.ws.take(work); S.xn = S.ws.take(xn); S.qkv = S.ws.take(qkv); S.obuf = S.ws.take(ob); S.hid = S.ws.take(hid);
        S.decA = S.ws.take(dec); S.decB = S.ws.take(dec); S.zpart = S.ws.take(zp); S.zpart_cap = zp;
        S.deemb = S.ws.take(deemb); S.rspec = S.ws.take(rspec); S.frames = S.ws.take(frames);
        S.stageA = S.ws.take(stage); S.stageB = S.ws.take(stage);
        S.codes_tmp = reinterpret_cast<long long*>(S.ws.take(codes));
        S.loss = S.ws.take(B); S.loss_terms = S.ws.take(lterms);
        S.tickets = reinterpret_cast<int*>(S.ws.take(WsFields::N_TICKETS));     // zero from the hipMemset above; every launch leaves them zero
        if (!S.loss) ESCX_FAIL(ESCX_ERR_STATE, "workspace sizing bug");
        S.shp = s;
    }
    h->n_sets = sets;
    h->cap_clips = Btotal;
    use_set(h, 0);
    return ESCX_OK;
}

extern "C" int escx_reserve(escx_handle h, int B, int L) {
    if (!h) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    if (L < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_samples must be positive");
    return reserve_frames(h, B, 1 + L / h->cfg.hop_length);
}

// ------------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------------
// Called at the start of every entry point, i.e. when no launch sequence of this handle holds a map pointer on the host side.  Dropping every
// map needs the kernels that read them to have finished.
static int trim_maps(escx_handle_s* h) {
    if (h->maps.size() < 768) return 0;
    ESCX_HIP(hipDeviceSynchronize());
    for (auto& kv : h->maps) (void)hipFree(kv.second);
    h->maps.clear();
    return 0;
}

int escx::check_ready(escx_handle_s* h) {
    if (!h) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    if (!h->finalized) ESCX_FAIL(ESCX_ERR_STATE, "parameters not finalised (call escx_finalize_params)");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) ESCX_FAIL(ESCX_ERR_HIP, "hipSetDevice failed");
    return trim_maps(h);
}

// Entry points that run the fp64-folded de-embedding of the inference path: after a device-side weight refresh (escx_train_forward with a flat
// buffer, escx_load_flat_params(full = 0)) that host-side product is out of date and the call would silently decode with the OLD weights.
int escx::check_infer_ready(escx_handle_s* h) {
    int rc = check_ready(h); if (rc) return rc;
    if (h->composed_stale)
        ESCX_FAIL(ESCX_ERR_STATE, "weights were refreshed on the device (training step): call escx_load_flat_params(handle, flat, /*full=*/1, stream) "
                                  "before decoding, so that the folded de-embedding is rebuilt from the current values");
    return 0;
}

int escx::ensure_ws(escx_handle_s* h, int B, int T, Shapes* s, int min_set_clips) {
    if (!ws_fits(h, B, T) || h->sets[0].shp.B < min_set_clips) {
        int rc = reserve_frames(h, B, T, min_set_clips);
        if (rc) return rc;
    }
    use_set(h, 0);
    return make_shapes(h, B, T, s);
}

