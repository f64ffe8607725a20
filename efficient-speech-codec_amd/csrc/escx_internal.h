// Internal host-side structures of libescx (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/escx.h"

namespace escx {

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

void set_error(const char* fmt, ...);

#define ESCX_FAIL(code, ...) do { ::escx::set_error(__VA_ARGS__); return (code); } while (0)
#define ESCX_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
        ::escx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return ESCX_ERR_HIP; } } while (0)

struct Param { std::vector<float> data; std::vector<int64_t> shape; };

struct Arena {                       // device bump allocator
    char* base = nullptr; size_t cap = 0, used = 0;
    float* take(size_t n_floats) {
        size_t bytes = (n_floats * sizeof(float) + 255) / 256 * 256;
        if (used + bytes > cap) return nullptr;
        float* p = reinterpret_cast<float*>(base + used); used += bytes; return p;
    }
};

struct BlockW {                      // one SwinBlock, packed
    float *ln1_g, *ln1_b, *wqkv, *bqkv, *bias_tab, *wproj, *bproj, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
    float *w1f, *w2f;                // fc1 / fc2 in MFMA fragment order ([n-tile][k-step][lane][4]) for the fused MLP
    float *wcf;                      // per hidden tile [fc1 fragments | fc2 fragments]: the LDS-staged fused MLP's stream
    float *waf, *baf, *bias_tab_f;   // fused attention: weight stream [group][tile][KK][64][4], tile biases, padded bias table
    float *wqkvT = nullptr, *wprojT = nullptr, *w1T = nullptr, *w2T = nullptr;     // transposed copies for the dX GEMMs of the training step
    long long tab_off = -1;          // flat offset of attn.relative_position_bias_table (its gradient is written there directly)
    bool x3a_pairs = false;          // x3a is in pair order (output projection split as well: fused_attn.h X3P)
    // Split-operand weight images (derived state, rebuilt by ensure_derived() in escx_api.cpp whenever the parameters or the precision mode change).  `x3a` / `x3w` are the
    // ACTIVE images (null = this block runs the fp32-MFMA kernel); the `_buf` pointers own the allocations, which are made once and kept across mode switches.
    void* x3a = nullptr;             // fused_attn.h X3: Q / K / V tiles split into terms
    void* x3w = nullptr;             // fused_mlp_x3.h: fc1 / fc2 split into terms
    void *x3a_buf = nullptr, *x3w_buf = nullptr;
    void* x3w_train = nullptr;       // training forward (train.hip): the fused MLP's weights split exactly into three bf16 terms, re-packed after every parameter refresh
};

struct Layer {                       // one TransformerLayer (attention.py:9-91)
    std::string prefix;
    int C, Cp, nH, hd, hdp, Nqkv, Ko, hidden, hiddenP;
    int attn_mode = -1, n_groups = 0; // fused attention head->tile mapping (fused_attn.h), -1 = not supported
    int scale;                       // 0 none, 1 down (PatchMerge), 2 up (PatchSplit)
    int Cout, CoutP;
    std::vector<BlockW> blocks;
    float *sub_g = nullptr, *sub_b = nullptr, *sub_w = nullptr;
    float *sub_wf = nullptr;         // scale-change weights in fragment order (fused LN + linear)
    void* sub_x3 = nullptr;          // the same, split into terms (rowgemm_x3_kernel; active image / owning buffer as BlockW::x3w)
    void* sub_x3s = nullptr;         // PatchSplit weights in the k-slot order of mlp_x3_kernel's SPLIT epilogue
    void *sub_x3_buf = nullptr, *sub_x3s_buf = nullptr;
    float *sub_wT = nullptr;         // transposed for dX (training)
};

struct Quant {                       // one ProductVectorQuantize (quantization.py:7-136)
    std::string prefix;
    int C, Cp, Hq, d, dt, Nz, Kq, Kup;
    float *wd, *cbn, *c2, *cbraw, *wup;
    float *wdT = nullptr, *wupT = nullptr;       // transposed for dX (training)
    // fused product-VQ kernel (fused_pvq.h): down-projection in MFMA fragment order ([k chunk][n tile][lane][4]: one coalesced 1 KiB fetch per
    // fragment), the group of every float4 of the framed vector in memory order (as floats; -1 = padding), and the de-quantisation TABLE
    //   tab[(h, ov * code + o)][c] = up_proj_g(codebook_g[code]) for the group g that owns element (o, h, c)
    // built on the device by pvq_up_kernel itself on 1024 pseudo-vectors (bit-identical to the up-projection it replaces; inference only,
    // rebuilt after a parameter refresh like the folded de-embedding)
    float *wdf = nullptr, *gq = nullptr, *tab = nullptr;
    bool tab_ok = false;
};

struct Shapes {                      // geometry for one (batch, n_samples)
    int B = 0, L = 0, T = 0, W = 0, H0 = 0, Tq = 0;
    std::vector<int> encH;           // H at encoder scale i (i = 0..n-1)
};

struct WsFields {                    // one workspace: every scratch buffer of the launch sequences
    Arena ws;
    Shapes shp;                      // shapes this set was sized for (shp.B = clips it can hold)
    float *spec = nullptr, *work = nullptr, *xn = nullptr, *qkv = nullptr, *obuf = nullptr, *hid = nullptr;
    float *decA = nullptr, *decB = nullptr, *zpart = nullptr, *deemb = nullptr, *rspec = nullptr, *frames = nullptr;
    float *stageA = nullptr, *stageB = nullptr, *loss = nullptr, *loss_terms = nullptr;     // loss_terms: [max_streams][G][B*Tq] per-vector commitment terms
    long long* codes_tmp = nullptr;
    int* tickets = nullptr;          // arrival counters of the in-launch combine of the hidden-split MLP (fused_mlp.h): zero between launches
    static constexpr int N_TICKETS = 16384;
    std::vector<float*> enc_hs;
    size_t zpart_cap = 0;
};

}  // namespace escx

struct escx_handle_s;
namespace escx {
int make_shapes(escx_handle_s* h, int B, int T, Shapes* out);
int check_infer_ready(escx_handle_s* h);     // check_ready + the folded inference layouts are current (see escx_api.cpp)
void free_train_state(escx_handle_s* h);     // train.hip: deletes the TrainTape bookkeeping object
int get_map(escx_handle_s* h, int H, int W, int shift, const int** out);     // shift 0/2: slot -> token; -1: merge rows; 10/12: token -> slot
int check_ready(escx_handle_s* h);
// Grow-only device scratch for the stateless entry points (losses, GAN terms), one buffer per (device, stream, slot): hipMallocAsync /
// hipFreeAsync cost ~1.6 ms of host time per call on this stack and were 40 % of the adversarial step's host time.  Work on one stream is
// ordered, so consecutive calls may reuse the buffer; growing synchronises the device once.
float* stream_scratch(hipStream_t st, int slot, size_t floats);
int launch_ok(const char* what);
int build_gather_map(escx_handle_s* h);
// escx_params.cpp, used by the launch sequences of escx_api.cpp
void use_set(escx_handle_s* h, int i);          // makes workspace set i the current one (the inherited WsFields)
int n_parts(escx_handle_s* h, int B);           // batch parts (streams) a batch of B clips runs as
int ensure_ws(escx_handle_s* h, int B, int T, Shapes* s, int min_set_clips = 0);      // reserve on demand (a set holds at least min_set_clips clips), select set 0, derive the shapes
int pass_clips(escx_handle_s* h, int B, int T);  // clips one pass of one part handles (<= clips of the part)
}

struct escx_handle_s : escx::WsFields {      // the inherited fields are the CURRENT set (swapped by use_set)
    escx_config cfg;
    int device = 0;
    int n = 0;                       // n_scales
    int F = 0, Fp = 0, n_fft = 0, left = 0, winP = 0, Kpe = 0, C0 = 0, C0p = 0, Q = 0;
    std::map<std::string, escx::Param> params;
    std::vector<std::string> required;
    bool finalized = false;
    bool use_fused = true;           // ESCX_NO_FUSED=1 selects the unfused GEMM pipeline (A/B and fallback)
    int mlp_variant = -1;            // ESCX_MLP_VARIANT overrides the per-layer choice (tuning)
    int attn_gs = 0;                 // ESCX_ATTN_GS: same for the head groups of the fused attention
    int mlp_hs = 0;                  // ESCX_MLP_HS: 0 = automatic hidden split, 1 = off, n = force n-way (tuning)
    int attn_nw = 0;                 // ESCX_ATTN_NW: waves per workgroup of the fused attention kernel (4 or 8)
    bool use_fused_attn = true;      // ESCX_NO_FUSED_ATTN=1
    // Arithmetic of the dense contractions with K = C (MLPs, Q / K / V, PatchMerge / PatchSplit, de-embedding): escx_set_precision (include/escx.h).
    //   0 = fp32 MFMA, 3 = three bf16 terms per fp32 operand (exact split), 2 = two fp16 terms (range rule of split_terms.h).  The environment only sets the DEFAULT.
    int prec = 2;
    int mlp_x3_max = 384, attn_x3_max = 384;     // ESCX_MLP_X3 / ESCX_ATTN_X3: largest padded width that runs split (A/B and fallback switches; 0 = that family on the fp32 MFMA)
    bool rowgemm_x3 = true, pvq_table = true;    // ESCX_ROWGEMM_X3=0 / ESCX_PVQ_TABLE=0
    // The remaining fallback / A-B switches of the launch sequences.  Round 6: EVERY product switch is a field of the handle, filled once by env_defaults() in escx_params.cpp -
    // the one place of the inference host code that reads the environment (train.hip / disc.hip keep their own few; tuning switches of rejected forms: tune_env.h).
    int attn_gs_tokens = 600;        // ESCX_ATTN_GS_TOKENS: head-group split of the attention for maps of up to this many tokens per clip (0 = off)
    bool mlp_split_fold = true;      // ESCX_MLP_SPLIT_FOLD=0: PatchSplit as its own launch instead of the MLP epilogue
    bool pvq_fused = true, pvq_up_kernel = true;      // ESCX_PVQ_FUSED=0: three-launch quantiser; ESCX_PVQ_UP_KERNEL=0: the GEMM engine's generic up-projection
    bool prof_serial = false;        // ESCX_PROF_SERIAL=1: profiled runs put the batch parts back to back
    int ws = 4;                      // window_size of the configuration (4: fused kernels; other sizes: unfused sequence with window_attention_any_kernel)
    bool attn_pack = true;           // ESCX_NO_ATTN_PACK=1: do not pack half-real windows of the H == 2 scale

    escx::Arena wts;                 // packed weights
    std::vector<escx::Layer> layers; // 2n entries (see escx_transformer_layer)
    std::vector<escx::Quant> quants; // max_streams entries
    float *pe_w = nullptr, *pe_b = nullptr, *pe_g = nullptr, *pe_beta = nullptr;
    float *dc1_w = nullptr, *dc1_b = nullptr, *dc2_w = nullptr, *dc2_b = nullptr;
    float *dft_w = nullptr, *idft_w = nullptr, *win2 = nullptr;
    float *dcc_w = nullptr, *dcc_b = nullptr, *dcv_w = nullptr, *dcv_b = nullptr;   // composed de-embedding: interior GEMM weights, border variants
    float* dch_w = nullptr;          // the interior weights as MFMA fragments for the halo-tiled kernel
    void* dch_x2 = nullptr;          // ... as the two-term fp16 stream of deembed7_x2_kernel (derived inference state, rebuilt with the tables; active pointer / owning buffer)
    void* dch_x2_buf = nullptr;
    float *dc1_wT = nullptr, *idft_wT = nullptr;     // training: conv5x5 dX weights, transposed inverse-DFT matrix

    // ---- training step (train.hip) ----
    std::vector<std::string> flat_keys;          // canonical flat order of the trainable parameters (== required keys)
    std::vector<size_t> flat_off, flat_numel;
    size_t flat_total = 0;
    int* gmap = nullptr;                         // per arena float: flat index + 1, 0 = zero padding, -1 = computed elsewhere
    float* garena = nullptr;                     // gradients of the packed layouts, same offsets as the weight arena
    std::vector<std::pair<size_t, size_t>> grad_regions;     // (offset, n) of the primary training layouts inside the arena
    long long* grad_seg = nullptr; int grad_nseg = 0; long long grad_seg_total = 0;     // device table [nseg][2] = (first element, arena offset) for the one-launch scatter
    escx::Arena tape;                            // activations kept between escx_train_forward and escx_train_backward
    void* train_state = nullptr;                 // TrainTape* (train.hip)
    bool train_x3_stale = true;                  // BlockW::x3w_train images are out of date (finalisation, device-side parameter refresh)
    bool pvq_tab_stale = true;                   // de-quantisation tables (Quant::tab) are out of date: rebuilt on the caller's stream by the next inference entry
    long long* iota_codes = nullptr;             // [G][Ksz] int64, codes[g][k] = k: the pseudo-vectors the tables are built from
    bool composed_stale = false;                 // weights were refreshed on the device: the fp64-folded de-embedding of the inference path is out of date
    bool deembed_halo = true;        // ESCX_DEEMBED_GEMM=1: implicit-GEMM form of the composed convolution instead (A/B, fallback)
    bool deembed_two_stage = false;  // ESCX_DEEMBED_TWO_STAGE=1: run conv5x5 and conv3x3 separately (A/B, fallback)

    // two workspace sets: whole-path calls split the batch in halves (clips are independent) and run them on two streams
    static constexpr int MAX_PARTS = 4;
    escx::WsFields sets[MAX_PARTS];
    int n_sets = 1;
    int cap_clips = 0;               // total clips (over all parts) the current workspace was reserved for
    int parts = 2;                   // ESCX_STREAMS=k (1..4): batch split into k parts on k streams; 1 = single stream
    bool parts_forced = false;       // ESCX_STREAMS given: use it for every batch size
    int chunk_frames = 10818;        // (18 clips of 3 s) a part walks its clips in passes of at most chunk_frames / T clips (0 = one pass): see run_halves (escx_api.cpp); ESCX_CHUNK_FRAMES
    hipStream_t sx[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};   // extra streams (part 0 runs on the caller's stream)
    hipEvent_t ev_fork = nullptr, ev_join[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};

    void* coll_buf = nullptr; size_t coll_cap = 0;   // int16 staging of escx_allgather_codes (send | recv)

    // index maps (device), keyed by (H, W, shift) ; shift = -1 -> merge map
    std::map<std::tuple<int, int, int>, int*> maps;

    // per-launch HIP-event profiler (escx_profile_*); off by default
    bool prof = false, prof_isolated = false;
    struct ProfRec { std::string name; double flops, bytes; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::string prof_json;
};

#include "launch_prof.h"
namespace escx {
hipEvent_t prof_event(escx_handle_s* h);
// One profiled launch group.  The group's first kernel launch through ESCX_LAUNCH (launch_prof.h: every launcher of the inference path) is made with
// hipExtLaunchKernel, so the pair (lt.start, lt.stop) brackets the DISPATCH ITSELF - begin and end of the kernel as rocprofv3's kernel trace sees them.  A group of
// several launches ends with a marker event after the last one; a group whose launches do not go through ESCX_LAUNCH (train.hip's own kernels) is bracketed by
// marker events as before; a group that launched nothing leaves a (near zero-length) record, which the callers that probe for an instantiation pop again.
struct ProfScope {
    escx_handle_s* h; hipStream_t st; LaunchTimer lt; LaunchTimer* prev = nullptr; hipEvent_t m_start = nullptr;
    ProfScope(escx_handle_s* h_, hipStream_t s) : h(h_), st(s) {
        if (h->prof) { m_start = prof_event(h); (void)hipEventRecord(m_start, st); lt.start = prof_event(h); lt.stop = prof_event(h); prev = g_launch_timer; g_launch_timer = &lt; }
    }
    ~ProfScope() { if (lt.start && g_launch_timer == &lt) g_launch_timer = prev; }
    void end(const std::string& name, double flops, double bytes) {
        if (!lt.start) return;
        g_launch_timer = prev;
        if (lt.launches != 1) (void)hipEventRecord(lt.stop, st);
        const bool dispatch_timed = lt.launches >= 1;
        h->prof_recs.push_back({name, flops, bytes, dispatch_timed ? lt.start : m_start, lt.stop});
        h->prof_pool.push_back(dispatch_timed ? m_start : lt.start);
        lt.start = nullptr;
    }
};
}  // namespace escx
#define PROF(name, flops, bytes, stmt) do { ::escx::ProfScope _ps(h, st); stmt; if (h->prof) _ps.end(name, flops, bytes); } while (0)
