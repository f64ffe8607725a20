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
extern "C" const char* escx_version(void) { return "escx 0.2 (gfx950, fp32 accumulate; fp32 MFMA + split-operand bf16 MFMA)"; }

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
    if (c.window_size != 4) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "only window_size=4 is implemented (got %d)", c.window_size);
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

extern "C" int escx_create(const escx_config* cfg, int device, escx_handle* out) {
    if (!cfg || !out) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null argument");
    int ndev = 0;
    ESCX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
    escx_handle_s* h = new escx_handle_s();
    h->cfg = *cfg; h->device = device;
    { const char* e = getenv("ESCX_NO_FUSED"); h->use_fused = !(e && e[0] == '1'); }
    { const char* e = ESCX_TUNE_ENV("ESCX_MLP_VARIANT"); if (e && e[0]) h->mlp_variant = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_MLP_HS"); if (e && e[0]) h->mlp_hs = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_GS"); if (e && e[0]) h->attn_gs = atoi(e); }
    { const char* e = getenv("ESCX_STREAMS"); if (e && e[0]) { h->parts = std::min(std::max(atoi(e), 1), (int)escx_handle_s::MAX_PARTS); h->parts_forced = true; } }
    { const char* e = getenv("ESCX_DEEMBED_TWO_STAGE"); h->deembed_two_stage = (e && e[0] == '1'); }
    { const char* e = getenv("ESCX_DEEMBED_GEMM"); h->deembed_halo = !(e && e[0] == '1'); }
    { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_NW"); if (e && e[0]) h->attn_nw = atoi(e); }
    { const char* e = ESCX_TUNE_ENV("ESCX_NO_ATTN_PACK"); h->attn_pack = !(e && e[0] == '1'); }
    { const char* e = getenv("ESCX_NO_FUSED_ATTN"); h->use_fused_attn = !(e && e[0] == '1'); }
    // precision DEFAULT of new handles (escx_set_precision changes it per handle): ESCX_PRECISION=fp32|bf16x3|f16x2 (or 0|3|2); ESCX_X3_TERMS=3 is the round-5 spelling of bf16x3
    { const char* e = getenv("ESCX_X3_TERMS"); if (e && atoi(e) == 3) h->prec = 3; }
    { const char* e = getenv("ESCX_PRECISION");
      if (e && e[0]) {
          const std::string v(e);
          if (v == "fp32" || v == "0") h->prec = 0; else if (v == "bf16x3" || v == "3") h->prec = 3; else if (v == "f16x2" || v == "2") h->prec = 2;
          else { delete h; ESCX_FAIL(ESCX_ERR_INVALID_ARG, "ESCX_PRECISION=%s (fp32 | bf16x3 | f16x2)", e); }
      } }
    { const char* e = getenv("ESCX_MLP_X3"); if (e && e[0]) h->mlp_x3_max = atoi(e); }
    { const char* e = getenv("ESCX_ATTN_X3"); if (e && e[0]) h->attn_x3_max = atoi(e); }
    { const char* e = getenv("ESCX_ROWGEMM_X3"); h->rowgemm_x3 = !(e && e[0] == '0'); }
    { const char* e = getenv("ESCX_PVQ_TABLE"); h->pvq_table = !(e && e[0] == '0'); }
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
    for (Layer& L : h->layers) for (BlockW& bw : L.blocks) { if (bw.x3w_buf) (void)hipFree(bw.x3w_buf); if (bw.x3a_buf) (void)hipFree(bw.x3a_buf); }
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
            GETP(tab, p + "attn.relative_position_bias_table", 49, nH);
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
            // relative position bias gathered per head: index = (dh+3)*7 + (dw+3)  (attention.py:195-205)
            o = slot(&bw.bias_tab, (size_t)nH * 256);
            for (int hh = 0; hh < nH; ++hh) for (int i = 0; i < 16; ++i) for (int jj = 0; jj < 16; ++jj) {
                const int idx = ((i >> 2) - (jj >> 2) + 3) * 7 + ((i & 3) - (jj & 3) + 3);
                pk.host[o + ((size_t)hh * 16 + i) * 16 + jj] = tab->data[(size_t)idx * nH + hh];
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
        const int sh0 = shift - 10;
        const int Hp = rup(H, 4), Wp = rup(W, 4), nWw = Wp / 4;
        m.assign((size_t)H * W, 0);
        for (int hh = 0; hh < Hp; ++hh) for (int ww = 0; ww < Wp; ++ww) {
            const int sh = (hh + sh0) % Hp, sw = (ww + sh0) % Wp;
            const int slot = (((hh >> 2) * nWw + (ww >> 2)) << 4) + ((hh & 3) << 2) + (ww & 3);
            if (sh < H && sw < W) m[(size_t)sh * W + sw] = slot;
        }
    } else
    if (shift >= 0) {                           // window slots -> source token (attention.py:139-155, 246-250)
        const int Hp = rup(H, 4), Wp = rup(W, 4), nWw = Wp / 4;
        m.resize((size_t)Hp * Wp);
        for (int hh = 0; hh < Hp; ++hh) for (int ww = 0; ww < Wp; ++ww) {
            const int sh = (hh + shift) % Hp, sw = (ww + shift) % Wp;       // roll(-shift) over the PADDED map
            const int slot = (((hh >> 2) * nWw + (ww >> 2)) << 4) + ((hh & 3) << 2) + (ww & 3);
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

static void use_set(escx_handle_s* h, int i) { static_cast<WsFields&>(*h) = h->sets[i]; }
// number of parts a batch of B clips is split into, and the clips one workspace set must hold
// Batches under 6 clips run as ONE part: splitting 2 - 4 clips over two streams costs more in per-launch efficiency than the overlap of the
// parts' tails returns (measured, tools/small_batch.py: B = 2 4.52 -> 4.12 ms, B = 4 4.96 -> 4.67 ms; from B = 8 up two parts win).  The
// arithmetic of a clip does not depend on how the batch is split, so the codes stay identical either way.
static int n_parts(escx_handle_s* h, int B) { return std::max(1, std::min((h->parts_forced || B >= 6) ? h->parts : 1, B)); }
static int set_clips(escx_handle_s* h, int B) { const int k = n_parts(h, B); return (B + k - 1) / k; }

static bool ws_fits(escx_handle_s* h, int B, int T) {
    const int need = set_clips(h, B), sets = n_parts(h, B);
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
    const int B = std::max(set_clips(h, Btotal), set_clips_min), sets = std::max(n_parts(h, Btotal), sets_min);
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
        if ((rc = get_map(h, H, s.W, 2, &dummy))) return rc;
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
        const size_t tokens = (size_t)B * H * s.W, slots = (size_t)B * rup(H, 4) * rup(s.W, 4);
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

static int ensure_ws(escx_handle_s* h, int B, int T, Shapes* s) {
    if (!ws_fits(h, B, T)) {
        int rc = reserve_frames(h, B, T);
        if (rc) return rc;
    }
    use_set(h, 0);
    return make_shapes(h, B, T, s);
}

// Whole-path calls: run `part(first_clip, n_clips, stream)` once, or as two halves on two streams joined by events.
// Clips are independent end to end, so the halves never exchange data; overlapping them lets one half's kernels fill
// the CUs the other half's tail workgroups leave idle (a 36-clip layer launches only ~1.3-2.6 workgroups per CU).
template <class F>
static int run_halves(escx_handle_s* h, int B, hipStream_t st, F part) {
    const int k = std::min(n_parts(h, B), h->n_sets);
    if (k <= 1) { use_set(h, 0); return part(0, B, st); }
    const int per = (B + k - 1) / k;
    // Event timing stays valid under concurrency (each kernel is bracketed on its own stream); ESCX_PROF_SERIAL=1 puts the
    // parts back to back on the caller's stream when isolated per-kernel durations are wanted.
    static const bool prof_serial = [] { const char* e = getenv("ESCX_PROF_SERIAL"); return e && e[0] == '1'; }();
    const bool concurrent = !(h->prof && (prof_serial || h->prof_isolated));
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
    for (int i = 0; i < k && !rc; ++i) {
        const int b0 = i * per, nb = std::min(per, B - b0);
        if (nb <= 0) break;
        const bool side = concurrent && i > 0;
        hipStream_t si = side ? h->sx[i] : st;
        if (side && hip_fail(hipStreamWaitEvent(si, h->ev_fork, 0), "fork wait")) break;
        use_set(h, i);
        if (side) started[i] = true;
        const int prc = part(b0, nb, si);
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

float* escx::stream_scratch(hipStream_t st, int slot, size_t floats) {
    struct Buf { float* p = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<std::tuple<int, hipStream_t, int>, Buf> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    Buf& b = bufs[std::make_tuple(dev, st, slot)];
    if (b.cap < floats) {
        (void)hipDeviceSynchronize();                                  // earlier work may still read the old buffer
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        const size_t want = floats + floats / 8 + 1024;
        if (hipMalloc((void**)&b.p, want * sizeof(float)) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.cap = want;
    }
    return b.p;
}

int escx::launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) ESCX_FAIL(ESCX_ERR_HIP, "%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// ---- per-launch profiler (ProfScope / PROF live in escx_internal.h) ---------------------------
thread_local escx::LaunchTimer* escx::g_launch_timer = nullptr;
hipEvent_t escx::prof_event(escx_handle_s* h) {
    if (!h->prof_pool.empty()) { hipEvent_t e = h->prof_pool.back(); h->prof_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

extern "C" int escx_profile_enable(escx_handle h, int enable) {
    if (!h) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    if (enable) { for (auto& r : h->prof_recs) { h->prof_pool.push_back(r.a); h->prof_pool.push_back(r.b); } h->prof_recs.clear(); }
    h->prof = enable != 0;
    h->prof_isolated = enable == 2;      // 2: run the batch parts back to back so that kernels do not share the GPU
    return ESCX_OK;
}

extern "C" const char* escx_profile_report(escx_handle h) {
    if (!h) return "[]";
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    struct Agg { int calls = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg; std::vector<std::string> order;
    for (auto& r : h->prof_recs) {
        float ms = 0.f; (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (!agg.count(r.name)) order.push_back(r.name);
        Agg& g = agg[r.name]; g.calls++; g.ms += ms; g.flops += r.flops; g.bytes += r.bytes;
    }
    std::string js = "[";
    char buf[512];
    for (size_t i = 0; i < order.size(); ++i) {
        const Agg& g = agg[order[i]];
        snprintf(buf, sizeof(buf), "%s{\"name\":\"%s\",\"calls\":%d,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}", i ? "," : "",
                 order[i].c_str(), g.calls, g.ms, g.flops, g.bytes);
        js += buf;
    }
    js += "]";
    h->prof_json = js;
    return h->prof_json.c_str();
}

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
static int attn_gs_for(int tokens_per_clip, int n_groups, int hiddenP, int Cp) { static const int lim = [] { const char* e = getenv("ESCX_ATTN_GS_TOKENS"); return e && e[0] ? atoi(e) : 600; }(); return (tokens_per_clip <= lim && n_groups % 3 == 0 && hiddenP >= 3 * Cp) ? 3 : 1; }
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
    const int Hp = rup(H, 4), Wp = rup(W, 4), slots = Hp * Wp, Ms = B * slots;
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
        const int shift = (j % 2 == 0) ? 0 : 2;                               // attention.py:29
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
            int gs = h->attn_gs > 0 ? (L.hiddenP >= h->attn_gs * L.Cp ? h->attn_gs : 1) : attn_gs_for(tokens, L.n_groups, L.hiddenP, L.Cp);
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
                static const bool split_fold_x3 = [] { const char* e = getenv("ESCX_MLP_SPLIT_FOLD"); return !(e && e[0] == '0'); }();
                if (split_fold_x3 && L.scale == 2 && j + 1 == L.blocks.size() && hs == 1 && L.sub_x3s) {      // PatchSplit in the epilogue of the layer's last MLP
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
            static const bool split_fold = [] { const char* e = getenv("ESCX_MLP_SPLIT_FOLD"); return !(e && e[0] == '0'); }();
            if (split_fold && L.scale == 2 && j + 1 == L.blocks.size() && hs == 1 && pend.n == 0) {
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
    static const bool special = [] { const char* e = getenv("ESCX_PVQ_UP_KERNEL"); return !(e && e[0] == '0'); }();     // 0: the GEMM engine's generic form (A/B, fallback)
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
    static const bool fused = [] { const char* e = getenv("ESCX_PVQ_FUSED"); return !(e && e[0] == '0'); }();
    if (fused) {
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
    rc = run_halves(h, B, (hipStream_t)stream, [&](int b0, int nb, hipStream_t st) -> int {
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
    rc = run_halves(h, B, (hipStream_t)stream, [&](int b0, int nb, hipStream_t st) -> int {
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
    return run_halves(h, B, (hipStream_t)stream, [&](int b0, int nb, hipStream_t st) -> int {
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
    int rc = ensure_ws(h, h->parts * B, frames_for_width(h, W), s);
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
