// Training step of ESC on MI355X: training-mode forward (activations kept), hand-written backward, losses, optimiser.
// Reference: esc/models/codecs.py:30-66 (forward), esc/models/csrvq.py:23-48,97-129 (cross-scale VQ in training mode),
// esc/modules/vq/codebook.py:57-75 (STE + losses), esc/modules/vq/quantization.py:31-72 (freeze_vq), scripts/trainer_no_adv.py:95-118 (the step),
// esc/modules/loss/generator_loss.py:12-74 (losses).  fp32 throughout, like the reference (no AMP there).
//
// Design: one stream, whole batch.  The forward is the plain GEMM pipeline with every tensor a backward needs kept on a "tape" (a bump
// arena in HBM: ~0.85 GB per 3 s clip for ESC-Base, 30 GB at batch 36 out of 288 GB).  The backward walks the tape in reverse; dX are
// gemm_engine launches with transposed weights, dW/db are gemm_dw launches whose partial sums are reduced in a fixed order.  Gradients
// land in `garena` (same offsets as the packed weight arena) and are scattered to the flat reference-layout gradient buffer at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "escx_internal.h"
#include "launchers.h"
#include "train_kernels.h"
#include "train_mlp_fused.h"
#include "gemm_bf16.h"

using namespace escx;

namespace {
// The training forward runs the FOLDED de-embedding (composed on the device by refresh_from_flat) when the geometry has one; the tape is sized
// without the fine map then.  ESCX_TRAIN_DEEMBED_COMPOSED=0: the reference's two convolutions (A/B baseline).
bool train_deembed_composed(const escx_handle_s* h) {
    static const bool want = [] { const char* e = getenv("ESCX_TRAIN_DEEMBED_COMPOSED"); return !(e && e[0] == '0'); }();
    return want && h->cfg.in_dim * h->Q <= 16 && h->dcv_w && h->dcv_b && h->dcc_w && h->dcc_b && h->dch_w;
}


struct BlockTape { float *x0, *xn1, *qkv, *obuf, *x1, *xn2, *hpre, *hact, *x2; };
struct LayerTape { std::vector<BlockTape> blk; float* sub_xn = nullptr; float* y = nullptr; int H = 0, Hout = 0; };
struct QuantTape { bool transmit = false; float *ze = nullptr, *zup = nullptr; const float* enc = nullptr; const float* dec = nullptr; float* out = nullptr; };
struct TrainTape {
    bool valid = false;
    long long generation = 0;                // counts escx_train_forward calls: a backward must belong to the LAST forward
    int B = 0, L = 0, S = 0, freeze = 0;
    Shapes shp;
    float *spec = nullptr, *pe_pre = nullptr, *tok0 = nullptr, *deemb = nullptr, *rspec = nullptr, *terms = nullptr;
    bool composed = false;                   // the forward ran the folded de-embedding: no fine map (deemb) on the tape
    long long* codes = nullptr;              // (B, max_streams, G, Tq)
    std::vector<LayerTape> layers;           // 2n
    std::vector<float*> enc_hs;              // n
    std::vector<QuantTape> q;                // max_streams
    float* post = nullptr;                   // decoder.post_nn output
    size_t fwd_mark = 0;
};

// Everything a training pass keeps between forward and backward.  Clips are independent in the training step too (per-clip losses, no batch statistics), so a
// large batch runs as TWO parts on two streams, like the inference path's run_halves: each part has its own tape arena, packed-gradient arena and flat
// gradient buffer; the parts' flat gradients are added in a fixed order after the join.  One part's launches fill the dispatch tails and small grids of the other.
struct TrainRoot {
    static constexpr int MAXP = 4;
    TrainTape single;                        // the one-part form (small batches, profiling, ESCX_TRAIN_PARTS=1)
    TrainTape part[MAXP];
    TrainTape* cur = &single;
    escx::Arena arena[MAXP];                 // tapes of the parts (the handle's own arena serves the one-part form)
    float* garena[MAXP] = {};                // [0] unused (part 0 writes the handle's)
    float* gflat[MAXP] = {};                 // [0] unused (part 0 writes the caller's buffer)
    hipStream_t aux[MAXP] = {}; hipEvent_t ev_fork = nullptr, ev_join[MAXP] = {};      // [0] unused (part 0 runs on the caller's stream)
    int parts = 1, first[MAXP + 1] = {};     // of the last forward: part p = clips [first[p], first[p + 1])
    bool valid = false;
    long long generation = 0;
};

}  // namespace
void escx::free_train_state(escx_handle_s* h) {
    TrainRoot* r = static_cast<TrainRoot*>(h->train_state);
    if (r) {
        for (int p = 0; p < TrainRoot::MAXP; ++p) {
            if (r->arena[p].base) (void)hipFree(r->arena[p].base);
            if (r->garena[p]) (void)hipFree(r->garena[p]);
            if (r->gflat[p]) (void)hipFree(r->gflat[p]);
            if (r->aux[p]) { (void)hipStreamSynchronize(r->aux[p]); (void)hipStreamDestroy(r->aux[p]); }
            if (r->ev_join[p]) (void)hipEventDestroy(r->ev_join[p]);
        }
        if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    }
    delete r;
    h->train_state = nullptr;
}
namespace {
TrainRoot* root_of(escx_handle_s* h) {
    if (!h->train_state) h->train_state = new TrainRoot();
    return static_cast<TrainRoot*>(h->train_state);
}
TrainTape* tape_of(escx_handle_s* h) { return root_of(h)->cur; }

inline unsigned blocks_for(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

void reduce_partials(const float* part, int slices, long long n, float* out, int accumulate, hipStream_t st) {
    launch_reduce_partials(part, slices, n, out, accumulate, st);
}

// ---- GEMM helpers --------------------------------------------------------------------------------
// Split-operand route of the linear layers (round 5): every fp32 operand as three exact bf16 terms, six cross products per tile on the bf16 MFMA, fp32
// accumulation (gemm_bf16.h NTERM = 3) - fp32-grade results at 1.4-1.7x the fp32 MFMA's rate where the tiles fill the chip.  ESCX_TRAIN_X3=0: the fp32 engine.
bool train_x3() {
    static const bool on = [] { const char* e = getenv("ESCX_TRAIN_X3"); return !(e && e[0] == '0'); }();
    return on;
}
inline bool train_x3_shape(int Np, int Kp) {          // by the layer's geometry only: the arithmetic of a clip must not depend on the batch it is in (two-part passes, DDP shards)
    static const int min_k = [] { const char* e = ESCX_TUNE_ENV("ESCX_TRAIN_X3_MIN_K"); return e ? atoi(e) : 96; }();       // tuning aid (tagged builds)
    return Kp % 16 == 0 && Kp >= min_k && Np >= min_k;
}

template <class Ld, class Epi>
void gemm_any(const Ld& ld, const float* W, int M, int Np, int Kp, const Epi& ep, hipStream_t st, int force_bk = 0) {
    if constexpr (std::is_same<Ld, PlainA>::value && !epi_is_rowwise<Epi>::value) {
        if (train_x3() && train_x3_shape(Np, Kp)) { launch_gemm_x3(PlainG{ld.A, ld.lda, ld.M, Kp}, W, M, Np, Kp, ep, st); return; }
    }
    // 128-row tiles halve the weight traffic per output row; 64 when the grid would not fill the chip (same rule as gemm_swin.hip)
    const long long tiles128 = (long long)((M + 127) / 128) * ((Np + 95) / 96);
    // K steps of 16 keep the workgroup's LDS image small (more resident workgroups): 98.0 -> 96.1 ms/step over the engine's default steps
    static const int env_bk = [] { const char* e = ESCX_TUNE_ENV("ESCX_TRAIN_BK"); return e ? atoi(e) : 16; }();
    if (!force_bk && env_bk > 0 && Kp % env_bk == 0) force_bk = env_bk;
    if (tiles128 >= 512) launch_gemm<128>(ld, W, M, Np, Kp, ep, st, 1, force_bk);
    else launch_gemm<64>(ld, W, M, Np, Kp, ep, st, 1, force_bk);
}
template <class Epi>
void gemm_rows(const float* A, int lda, int M, const float* W, int Np, int Kp, const Epi& ep, hipStream_t st) {
    gemm_any(PlainA{A, lda, M}, W, M, Np, Kp, ep, st);
}

struct Scratch {                              // bump allocator over the free tail of the tape
    Arena* a; size_t mark;
    explicit Scratch(Arena* ar) : a(ar), mark(ar->used) {}
    ~Scratch() { a->used = mark; }
    float* take(size_t n) { return a->take(n); }
};

constexpr size_t DW_PART_FLOATS = (size_t)40 << 20;      // partial-sum scratch of one dW launch (160 MB)

// dW[Np][Kp] (+ db[Np]) = sum over M rows; la: gradient rows (columns n), lb: saved input rows (columns k)
template <class LdA, class LdB>
int dw_launch(escx_handle_s* h, const LdA& la, const LdB& lb, int M, int Np, int Kp, float* dW, float* db, float* part, hipStream_t st) {
    const int nbn = (Np + 47) / 48, nbk = (Kp + 47) / 48, blocks = nbn * nbk;
    // Workgroups per launch.  A workgroup's fixed cost (cold first chunk, cross-wave add of 9 accumulator tiles, 9 KB partial tile that the reduction reads
    // back) is paid per slice, so the narrow matrices - few output tiles, i.e. many slices each - want FEWER, longer slices even if that leaves slots empty
    // (measured, B = 36, ESCX_DW_TARGET sweep in profiles/r3_dw_target_sweep.txt: dw_proj[C=45] (1 tile) 0.631 / 0.535 / 0.444 ms at 2048 / 1024 / 512,
    // dw_qkv[C=45] (3 tiles) 1.218 / 1.137 / 1.026; from 10 tiles up 2048 wins: dw_fc1[C=72] 1.275 / 1.270 / 1.793).  ESCX_DW_TARGET overrides (tuning aid).
    static const int env_target = [] { const char* e = ESCX_TUNE_ENV("ESCX_DW_TARGET"); return e ? atoi(e) : 0; }();
    const int target = env_target > 0 ? env_target : (blocks <= 3 ? 512 : (blocks <= 4 ? 1024 : 2048));
    int slices = std::max(1, std::min((target + blocks - 1) / blocks, (M + 127) / 128));
    const size_t per = (size_t)Np * Kp + Np;
    slices = (int)std::min<size_t>(slices, DW_PART_FLOATS / per);
    if (slices < 1) ESCX_FAIL(ESCX_ERR_STATE, "dW scratch too small for %d x %d", Np, Kp);
    int mps = ((M + slices - 1) / slices + 127) / 128 * 128;
    slices = (M + mps - 1) / mps;
    float* bpart = part + (size_t)slices * Np * Kp;
    if (db) hipLaunchKernelGGL((gemm_dw_kernel<LdA, LdB, true>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
    else hipLaunchKernelGGL((gemm_dw_kernel<LdA, LdB, false>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
    reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
    if (db) reduce_partials(bpart, slices, (long long)Np, db, 0, st);
    return 0;
}

// the same through wide workgroup tiles (gemm_dw3_kernel, 2 x 2 waves of TA x TB accumulator tiles): operands staged once per workgroup
template <int TA, int TB, int WN = 2, int WK = 2, class LdA, class LdB>
int dw_launch_wide(escx_handle_s* h, const LdA& la, const LdB& lb, int M, int Np, int Kp, float* dW, float* db, float* part, hipStream_t st) {
    constexpr int WA = 16 * TA * WN, WB = 16 * TB * WK;
    const int nbn = (Np + WA - 1) / WA, nbk = (Kp + WB - 1) / WB, blocks = nbn * nbk;
    static const int env_target = [] { const char* e = ESCX_TUNE_ENV("ESCX_DW_WIDE_TARGET"); return e ? atoi(e) : 0; }();
    // ONE round of resident workgroups (2 per CU at 65-74 KB of LDS each): a partly filled extra round costs a whole slice time, and every further slice
    // another partial tile (up to 64 KB) written and read back (ESCX_DW_WIDE_TARGET sweep, 36 clips: 512 -> 79.6 ms/step, 768 -> 80.9, 2560 -> 80.4)
    constexpr int LDS_BYTES = 2 * 32 * ((WA % 32 == 0 ? WA + 16 : WA) + (WB % 32 == 0 ? WB + 16 : WB)) * 4;
    const int target = env_target > 0 ? env_target : 256 * std::max(1, std::min(2, 163840 / LDS_BYTES));
    int slices = std::max(1, std::min(target / blocks, (M + 255) / 256));
    const size_t per = (size_t)Np * Kp + Np;
    slices = (int)std::max<size_t>(1, std::min<size_t>(slices, DW_PART_FLOATS / per));
    int mps = ((M + slices - 1) / slices + 31) / 32 * 32;
    slices = (M + mps - 1) / mps;
    float* bpart = part + (size_t)slices * Np * Kp;
    if (db) hipLaunchKernelGGL((gemm_dw3_kernel<LdA, LdB, TA, TB, WN, WK, true>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
    else hipLaunchKernelGGL((gemm_dw3_kernel<LdA, LdB, TA, TB, WN, WK, false>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, bpart);
    reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
    if (db) reduce_partials(bpart, slices, (long long)Np, db, 0, st);
    return 0;
}

// plain row-major operands (the linear layers): the workgroup tile (128 / 96 / 64 per side) with the least padding, wide tiles preferred;
// the 48 x 48 kernel where none fits within 15 % (C = 45, 72, 144: multiples of 48) or the matrix is small
int dw_rows(escx_handle_s* h, const float* A, int lda, const float* Bm, int ldb, int M, int Np, int Kp, float* dW, float* db, float* part, hipStream_t st) {
    static const bool wide_ok = [] { const char* e = ESCX_TUNE_ENV("ESCX_DW_WIDE"); return !(e && e[0] == '0'); }();
    static const double pad_limit = [] { const char* e = ESCX_TUNE_ENV("ESCX_DW_WIDE_PAD"); return e ? atof(e) : 1.15; }();      // tile padding a wide tile may add
    const PlainA la{A, lda, M}, lb{Bm, ldb, M};
    // split-operand dW (gemm_bf16.h gemm_dw_bf16_kernel<.., 3>: 128 x 128 tiles of dW, both operands as three bf16 terms, slices of M added in a fixed order) where the
    // tiles carry no padding - the C = 384 blocks; the narrower matrices keep the fp32 kernels with their fitted tile shapes below
    static const bool dw_x3 = [] { const char* e = ESCX_TUNE_ENV("ESCX_TRAIN_X3_DW"); return !(e && e[0] == '0'); }();
    if (train_x3() && dw_x3 && Np % 128 == 0 && Kp % 128 == 0) {
        const int nbn = Np / 128, nbk = Kp / 128, blocks = nbn * nbk;
        const size_t per = (size_t)Np * Kp + Np;
        int slices = std::max(1, std::min((1024 + blocks / 2) / blocks, (M + 255) / 256));
        slices = (int)std::max<size_t>(1, std::min<size_t>(slices, DW_PART_FLOATS / per));
        int mps = ((M + slices - 1) / slices + 31) / 32 * 32;
        slices = (M + mps - 1) / mps;
        float* bpart = part + (size_t)slices * Np * Kp;
        hipLaunchKernelGGL((gemm_dw_bf16_kernel<PlainA, PlainA, 3>), dim3(blocks, slices), dim3(256), 0, st, la, lb, M, Np, Kp, nbk, mps, part, db ? bpart : nullptr);
        reduce_partials(part, slices, (long long)Np * Kp, dW, 0, st);
        if (db) reduce_partials(bpart, slices, (long long)Np, db, 0, st);
        return 0;
    }
    // the 144- and 80-wide maps (C = 144, 72) fit none of the 32-multiple tiles: one side of the workgroup tile IS the map width (9 or 5 accumulator tiles per
    // wave, all four waves along the other side).  ESCX_DW_ODD=0: the 48 x 48 kernel as before.
    static const bool odd_ok = [] { const char* e = ESCX_TUNE_ENV("ESCX_DW_ODD"); return !(e && e[0] == '0'); }();
    if (wide_ok && odd_ok) {
        if (Kp == 144 && Np >= 256) return dw_launch_wide<2, 9, 4, 1>(h, la, lb, M, Np, Kp, dW, db, part, st);         // 128 x 144
        if (Np == 144 && Kp >= 256) return dw_launch_wide<9, 2, 1, 4>(h, la, lb, M, Np, Kp, dW, db, part, st);         // 144 x 128
        if (Kp == 80 && Np > 192 && Np <= 256) return dw_launch_wide<4, 5, 4, 1>(h, la, lb, M, Np, Kp, dW, db, part, st);      // 256 x 80: the whole QKV gradient of C = 72
    }
    int bestA = 0, bestB = 0; double best = 1e30;
    if (wide_ok && Np >= 96 && Kp >= 96 && (long long)Np * Kp >= 96 * 288) {
        const int cand[3] = {128, 96, 64};
        for (int wa : cand) for (int wb : cand) {
            const double padded = (double)((Np + wa - 1) / wa * wa) * ((Kp + wb - 1) / wb * wb);
            if (padded > pad_limit * Np * Kp) continue;
            const double cost = padded * (1.0 + 24.0 / wa + 24.0 / wb);          // MFMA work + a charge for operand re-staging
            if (cost < best) { best = cost; bestA = wa; bestB = wb; }
        }
    }
    switch (bestA * 1000 + bestB) {
        case 128128: return dw_launch_wide<4, 4>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 128096: return dw_launch_wide<4, 3>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 128064: return dw_launch_wide<4, 2>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 96128: return dw_launch_wide<3, 4>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 96096: return dw_launch_wide<3, 3>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 96064: return dw_launch_wide<3, 2>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 64128: return dw_launch_wide<2, 4>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 64096: return dw_launch_wide<2, 3>(h, la, lb, M, Np, Kp, dW, db, part, st);
        case 64064: return dw_launch_wide<2, 2>(h, la, lb, M, Np, Kp, dW, db, part, st);
        default: return dw_launch(h, la, lb, M, Np, Kp, dW, db, part, st);
    }
}

constexpr int LN_BWD_MAX_GRID = 2048;
// LayerNorm backward launcher; dgamma -> dg[SEGS*Cp], dbeta -> dbt[SEGS*Cp]
int ln_bwd(int mode, const float* x, const float* dy, const float* gamma, const int* map, const float* add, float* dx, float* dg, float* dbt,
           int rows_per_clip, int src_rows_per_clip, int dy_rows_per_clip, int total_rows, int C, int Cp, float* part, hipStream_t st,
           float* dx_slots = nullptr, const int* slot_of = nullptr, int slots_per_clip = 0) {
    const int segs = mode == 2 ? 2 : 1;
    static const int grid_cap = [] { const char* e = ESCX_TUNE_ENV("ESCX_LN_BWD_GRID"); const int v = e ? atoi(e) : 512; return std::max(1, std::min(v, LN_BWD_MAX_GRID)); }();
    const int grid = (int)std::min<long long>(grid_cap, ((long long)total_rows + 15) / 16);
    const size_t shm = (size_t)16 * 2 * segs * Cp * sizeof(float);
    const int RW = segs * Cp;
    if (mode == 0) hipLaunchKernelGGL((ln_bwd_kernel<1, 0>), dim3(grid), dim3(256), shm, st, x, dy, gamma, map, add, dx, part, rows_per_clip, src_rows_per_clip, dy_rows_per_clip, total_rows, C, Cp, 1e-5f, dx_slots, slot_of, slots_per_clip);
    else if (mode == 1) hipLaunchKernelGGL((ln_bwd_kernel<1, 1>), dim3(grid), dim3(256), shm, st, x, dy, gamma, map, add, dx, part, rows_per_clip, src_rows_per_clip, dy_rows_per_clip, total_rows, C, Cp, 1e-5f, dx_slots, slot_of, slots_per_clip);
    else hipLaunchKernelGGL((ln_bwd_kernel<2, 2>), dim3(grid), dim3(256), shm, st, x, dy, gamma, map, add, dx, part, rows_per_clip, src_rows_per_clip, dy_rows_per_clip, total_rows, C, Cp, 1e-5f, dx_slots, slot_of, slots_per_clip);
    // part: [grid][2][RW] -> reduce over grid (fixed order) into a [2][RW] row, then to the two destinations
    float* red = part + (size_t)grid * 2 * RW;
    reduce_partials(part, grid, (long long)2 * RW, red, 0, st);
    hipLaunchKernelGGL(copy2_kernel, dim3(blocks_for(2 * RW)), dim3(256), 0, st, red, dg, dbt, RW);          // one launch instead of two copy-engine packets (~7 us each)
    return 0;
}
constexpr size_t LN_PART_FLOATS = (size_t)(LN_BWD_MAX_GRID + 1) * 2 * 2 * 384;

// dx_fc1 GEMM with the LayerNorm backward as its row epilogue (EpiLnBwdRows; Cp <= 96): out = add + LNbwd(dh . W; x), dgamma / dbeta.
// `part` needs ceil(M / 64) * 4 * 2 * Cp floats (+ 2 * Cp for the reduced row): the dW partial-sum scratch is used.
// Round 4: also for the wide maps (Cp = 144 / 192 / 384) - one workgroup tile spans the row there too (BN = Cp), with the register-lean form of the epilogue.
bool ln_rows_fusable(int Cp) {
    static const bool wide = [] { const char* e = ESCX_TUNE_ENV("ESCX_LN_FUSED_WIDE"); return e && e[0] == '1'; }();          // opt-in: measured slower than the stand-alone LayerNorm backward above Cp = 96
    return Cp <= 96 || (wide && (Cp == 144 || Cp == 192 || Cp == 384));
}
void gemm_ln_bwd_rows(const float* A, int lda, int M, const float* Wt, int Cp, int Kp, const float* x, const float* gamma, const float* add, float* dx,
                      float* dx_slots, const int* slot_of, int rows_per_clip, int slots_per_clip, int C, float* dg, float* dbt, float* part, hipStream_t st,
                      const int* row_map = nullptr) {
    EpiLnBwdRows ep{x, gamma, add, dx, dx_slots, slot_of, part, C, Cp, rows_per_clip, slots_per_clip, 1e-5f, row_map};
    static const int wide_bm = [] { const char* e = ESCX_TUNE_ENV("ESCX_LNBWD_WIDE_BM"); return e ? atoi(e) : 0; }();       // tuning aid: 64 / 128 rows per workgroup for the 144 / 192-wide tiles
    const bool big = Cp > 96 ? (Cp != 384 && (wide_bm ? wide_bm == 128 : (long long)((M + 127) / 128) >= 512)) : (long long)((M + 127) / 128) >= 512;
    const int rows = (big ? (M + 127) / 128 : (M + 63) / 64) * 4;
    static const int env_bk = [] { const char* e = ESCX_TUNE_ENV("ESCX_LNBWD_BK"); return e ? atoi(e) : 16; }();
    const int bk = (env_bk > 0 && Kp % env_bk == 0) ? env_bk : 16;
    const PlainA la{A, lda, M};
    if (Cp == 144) { if (big) launch_tile<128, 144, 16>(la, Wt, M, Cp, Kp, 1, ep, st); else launch_tile<64, 144, 16>(la, Wt, M, Cp, Kp, 1, ep, st); }
    else if (Cp == 192) { if (big) launch_tile<128, 192, 16>(la, Wt, M, Cp, Kp, 1, ep, st); else launch_tile<64, 192, 16>(la, Wt, M, Cp, Kp, 1, ep, st); }
    else if (Cp == 384) launch_tile<64, 384, 16>(la, Wt, M, Cp, Kp, 1, ep, st);
    else if (big) launch_gemm<128>(la, Wt, M, Cp, Kp, ep, st, 1, bk);
    else launch_gemm<64>(la, Wt, M, Cp, Kp, ep, st, 1, bk);
    float* red = part + (size_t)rows * 2 * Cp;
    launch_reduce_partials(part, rows, (long long)2 * Cp, red, 0, st, red + 2 * Cp);
    hipLaunchKernelGGL(copy2_kernel, dim3(blocks_for(2 * Cp)), dim3(256), 0, st, red, dg, dbt, Cp);
}

int attn_bwd(const float* qkv, const float* bias, const float* dout, float* dqkv, float* dbias, float* part, int total_windows, int nH, int hdp, int ldq,
             int ldo, int nWh, int nWw, int shifted, float scale, hipStream_t st) {
    // Persistent waves striding over the windows, each doing HPW heads per window (the heads of a window share its rows: in one wave every cache line is
    // fetched once).  Grid: ONE workgroup per CU in total - measured at 36 clips (profiles/r3_attn_bwd_grid_sweep.txt): the kernel moves 1.06 GB per launch
    // at C = 45 and runs at 5.0 TB/s with 256 workgroups, 4.1 TB/s with 384 and 3.0 TB/s with the chip full (1 536, one head per wave): more concurrent row
    // streams only cost DRAM efficiency, there is no latency left to hide.  ESCX_ATTN_BWD_GX = n: n window chunks; ESCX_ATTN_BWD_HPW=1: one head per wave.
    static const int gx_env = [] { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_BWD_GX"); return e ? atoi(e) : 0; }();
    static const int hpw_env = [] { const char* e = ESCX_TUNE_ENV("ESCX_ATTN_BWD_HPW"); return e ? atoi(e) : 3; }();
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int hpw = (hpw_env == 3 && nH % 3 == 0) ? 3 : 1;
    const int gy = nH / hpw;
    int gx = (int)std::min<long long>(512, ((long long)total_windows + 3) / 4);
    if (gx_env > 0) gx = std::min(gx, gx_env);
    else {
        int cap = std::max(1, cus / gy);
        if (cap >= 16) cap &= ~7;                            // the chunks of one window range land on the same XCD for every head group (workgroup id mod 8)
        gx = std::min(gx, cap);
    }
    auto launch = [&](auto kern) {
        if ((unsigned long long)(total_windows + 4 * gx) * 16ull * (unsigned long long)std::max(ldq, ldo) >= (1ull << 32)) return -2;    // 32-bit element offsets in the kernel
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), 0, st, qkv, bias, dout, dqkv, part, total_windows, nH, ldq, ldo, nWh, nWw, shifted, scale);
        return 0;
    };
    int lrc = -1;
#define ESCX_ATB(S) case S: lrc = hpw == 3 ? launch(attn_bwd_kernel<S, 3>) : launch(attn_bwd_kernel<S, 1>); break;
    switch (hdp / 4) {
        ESCX_ATB(1) ESCX_ATB(2) ESCX_ATB(3) ESCX_ATB(4) ESCX_ATB(5) ESCX_ATB(6) ESCX_ATB(7) ESCX_ATB(8) ESCX_ATB(12) ESCX_ATB(16)
        default: return -1;
    }
    if (lrc) return lrc;
#undef ESCX_ATB
    reduce_partials(part, gx, (long long)nH * 256, dbias, 0, st);
    return 0;
}
constexpr size_t ATT_PART_FLOATS = (size_t)512 * 64 * 256;

inline float* G(escx_handle_s* h, const float* w) { return h->garena + (w - reinterpret_cast<const float*>(h->wts.base)); }

// The wide token maps (C = 45: 12 hidden tiles = the 12 waves of one workgroup) run the MLP fused in both directions: forward = the inference
// path's kernel (only x1 stays on the tape), backward = train_mlp_fused.h (hidden tile recomputed on the fly).  ESCX_TRAIN_MLP_FUSED=0: unfused.
inline bool mlp_train_fused(const Layer& L) {
    static const bool on = [] { const char* e = getenv("ESCX_TRAIN_MLP_FUSED"); return !(e && e[0] == '0'); }();
    static const bool on72 = [] { const char* e = ESCX_TUNE_ENV("ESCX_TRAIN_MLP_FUSED_C72"); return !(e && e[0] == '0'); }();
    return on && ((L.Cp == 48 && L.hiddenP == 192) || (on72 && L.Cp == 80 && L.hiddenP == 288));
}

// Forward of those layers (round 6): the split-operand kernel of the inference path (fused_mlp_x3.h) in its EXACT three-term bf16 form - fp32 results up to summation
// order, no range rule needed - on weight images re-packed after every parameter refresh (pack_train_mlp_images, 10 launches per step for ESC-Base).  The backward
// recomputes the hidden tile from x1 on the fp32 MFMA: the two evaluations of h_pre differ by fp32 rounding (~1e-7), which moves the gradient like any re-association.
// ESCX_TRAIN_MLP_X3=0: the fp32-MFMA fused kernel (fused_mlp.h), as before.
inline bool mlp_train_x3() {
    const char* e = getenv("ESCX_TRAIN_MLP_X3");                 // read per call: tests switch it
    return !(e && e[0] == '0');
}
int pack_train_mlp_images(escx_handle_s* h, hipStream_t st) {
    if (!mlp_train_x3() || !h->train_x3_stale) return 0;
    for (Layer& L : h->layers) {
        if (!mlp_train_fused(L)) continue;
        for (BlockW& bw : L.blocks) {
            if (!bw.x3w_train) ESCX_HIP(hipMalloc(&bw.x3w_train, mlp_x3_bytes(L.Cp, L.hiddenP, 3)));
            if (mlp_x3_pack(bw.w1, bw.w2, bw.x3w_train, L.Cp, L.hiddenP, st, 3, nullptr, nullptr, nullptr, L.C) != 0)
                ESCX_FAIL(ESCX_ERR_STATE, "split MLP image for Cp = %d could not be packed", L.Cp);
        }
    }
    h->train_x3_stale = false;
    return launch_ok("pack_train_mlp_images");
}

// One workgroup per CU (~150 KB of LDS), persistent over the row tiles.  C = 45: 12 hidden tiles = 12 compute waves, LayerNorm backward inside.
// C = 72: 18 hidden tiles as 2 x 9 (grid.y = 2): the workgroups write d xn partial slabs (`slabs`: 2 * M * Cp floats), which are summed into `dxn`
// and go through the stand-alone LayerNorm backward.  `part`: grid.x x (2 * hiddenP * Cp + hiddenP + Cp) floats + the reduced E.
int mlp_bwd_fused(escx_handle_s* h, const Layer& L, const BlockW& bw, const float* x1, const float* dy, float* dx1, float* dx1s, const int* slot_of,
                  int tokens, int slots, int M, float* part, float* dxn, float* slabs, float* lnpart, hipStream_t st) {
    const int ntiles = (M + 15) / 16;
    const bool split = L.Cp == 80;
    const int grid = std::min(ntiles, split ? 128 : 256);
    const int n1 = L.hiddenP * L.Cp;
    const size_t per = 2 * (size_t)n1 + L.hiddenP + L.Cp;
    if ((size_t)grid * per + n1 > DW_PART_FLOATS) ESCX_FAIL(ESCX_ERR_STATE, "dW scratch too small for the fused MLP backward");
    MlpBwdArgs a{x1, dy, dx1, dx1s, slot_of, bw.ln2_g, bw.ln2_b, bw.w1, bw.b1, bw.w2T, bw.w1T, part, slabs, M, L.C, L.hiddenP, tokens, slots, 1e-5f, 0};
    { static const int dbg = [] { const char* e = ESCX_TUNE_ENV("ESCX_MLPBWD_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
    // two-term fp16 form of the two channel contractions (train_mlp_fused.h X2; ESCX_TRAIN_MLPBWD_X2=0: all five contractions on the fp32 MFMA, as before)
    const char* x2e = getenv("ESCX_TRAIN_MLPBWD_X2");            // read per call: tests switch it
    const bool x2 = !(x2e && x2e[0] == '0');
    if (split) { if (x2) hipLaunchKernelGGL((mlp_bwd_fused_kernel<80, 9, 2, true>), dim3(grid, 2), dim3(64 * 12), 0, st, a); else hipLaunchKernelGGL((mlp_bwd_fused_kernel<80, 9, 2>), dim3(grid, 2), dim3(64 * 12), 0, st, a); }
    else { if (x2) hipLaunchKernelGGL((mlp_bwd_fused_kernel<48, 12, 1, true>), dim3(grid), dim3(64 * 15), 0, st, a); else hipLaunchKernelGGL((mlp_bwd_fused_kernel<48, 12, 1>), dim3(grid), dim3(64 * 15), 0, st, a); }
    float* Etot = part + (size_t)grid * per;
    hipLaunchKernelGGL(mlp_bwd_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, part, grid, n1, L.hiddenP, L.Cp, Etot, G(h, bw.w2),
                       G(h, bw.b1), G(h, bw.b2));
    hipLaunchKernelGGL(mlp_bwd_finish_kernel, dim3(1), dim3(1024), 0, st, Etot, G(h, bw.b1), bw.w1, bw.ln2_g, bw.ln2_b, G(h, bw.w1),
                       split ? nullptr : G(h, bw.ln2_g), split ? nullptr : G(h, bw.ln2_b), L.hiddenP, L.Cp);
    if (split) {
        const long long n4 = (long long)M * L.Cp / 4;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, slabs, 2, n4, dxn);
        ln_bwd(0, x1, dxn, bw.ln2_g, nullptr, dy, dx1, G(h, bw.ln2_g), G(h, bw.ln2_b), tokens, tokens, tokens, M, L.C, L.Cp, lnpart, st, dx1s, slot_of, slots);
    }
    return 0;
}

// ---- tape sizing -----------------------------------------------------------------------------------
int layer_H(escx_handle_s* h, const Shapes& s, int li) {
    const int n = h->n;
    if (li < n) return s.encH[std::max(li - 1, 0)];
    if (li < 2 * n - 1) return s.encH[n - 1 - (li - n)];
    return s.encH[0];
}

size_t pad256(size_t nfl) { return (nfl * sizeof(float) + 255) / 256 * 256; }

size_t tape_bytes(escx_handle_s* h, const Shapes& s) {
    const escx_config& c = h->cfg;
    const int n = h->n, B = s.B;
    size_t tot = 0, act_max = 0, hid_max = 0, qkv_max = 0;
    auto add = [&](size_t nfl) { tot += pad256(nfl); };
    add((size_t)B * s.T * c.in_dim * h->Fp);                         // spec
    const size_t tok0 = (size_t)B * s.H0 * s.W * h->C0p;
    add(tok0); add(tok0);                                            // pe_pre, tok0
    for (int li = 0; li < 2 * n; ++li) {
        const Layer& L = h->layers[li];
        const int H = layer_H(h, s, li);
        const size_t M = (size_t)B * H * s.W, Ms = (size_t)B * rup(H, 4) * rup(s.W, 4);
        for (size_t j = 0; j < L.blocks.size(); ++j) {
            add(Ms * L.Cp); add(Ms * L.Nqkv); add(Ms * L.Ko); add(M * L.Cp); add(M * L.Cp);
            if (!mlp_train_fused(L)) { add(M * L.Cp); add(M * L.hiddenP); add(M * L.hiddenP); }       // xn2, h_pre, gelu(h_pre): not kept by the fused MLP
        }
        if (L.scale == 1) { const size_t M2 = (size_t)B * ((H + 1) / 2) * s.W; add(M2 * 2 * L.Cp); add(M2 * L.CoutP); }
        else if (L.scale == 2) { add(M * L.Cp); add(2 * M * L.CoutP); }
        act_max = std::max({act_max, M * L.Cp, Ms * L.Cp, Ms * L.Ko, 2 * M * L.CoutP});
        hid_max = std::max(hid_max, M * L.hiddenP);
        qkv_max = std::max(qkv_max, Ms * L.Nqkv);
    }
    const size_t Mq = (size_t)B * s.Tq;
    size_t zp_max = 0;
    for (const Quant& q : h->quants) {
        add(Mq * q.Nz); add(Mq * q.Nz);                              // ze, zup
        add((size_t)B * q.Hq * s.W * q.Cp);                          // refined decoder map
        zp_max = std::max(zp_max, (size_t)pvq_down_splits((int)Mq, q.Kq, q.Cp) * Mq * q.Nz);
    }
    add((size_t)c.max_streams * c.group_size * Mq);                  // loss terms
    add((size_t)B * c.max_streams * c.group_size * s.Tq * 2);        // codes (int64)
    const int T2 = c.patch_t * s.W, F2 = c.patch_f * s.H0;
    if (!train_deembed_composed(h)) add((size_t)B * T2 * F2 * h->C0p);       // de-embedding fine map (two-convolution form only)
    add((size_t)B * T2 * c.in_dim * h->Fp);                          // rspec
    // backward scratch (upper bounds): activations-sized gradients, dW partials, LN / attention partials, frames
    const size_t fine = (size_t)B * T2 * F2 * h->C0p;
    tot += 12 * pad256(act_max) + 2 * pad256(hid_max) + 2 * pad256(qkv_max) + pad256(zp_max) + 2 * pad256(fine) + 2 * pad256(DW_PART_FLOATS) +
           pad256(LN_PART_FLOATS) + pad256(ATT_PART_FLOATS) + 3 * pad256((size_t)B * T2 * std::max(h->winP, c.in_dim * h->Fp)) + (size_t)n * pad256(act_max) + (64 << 20);
    return tot;
}

// ---- forward pieces -----------------------------------------------------------------------------------
int layer_fwd(escx_handle_s* h, const Layer& L, LayerTape& LT, const float* x_in, int B, int H, int W, hipStream_t st) {
    Arena& tp = h->tape;
    const int tokens = H * W, M = B * tokens;
    const int Hp = rup(H, 4), Wp = rup(W, 4), slots = Hp * Wp, Ms = B * slots;
    const double dM = M, dMs = Ms, dC = L.C, f4 = 4;
    LT.H = H;
    LT.blk.assign(L.blocks.size(), BlockTape());
    const std::string tg = h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string();
    const float* x = x_in;
    int rc;
    for (size_t j = 0; j < L.blocks.size(); ++j) {
        const BlockW& bw = L.blocks[j];
        BlockTape& bt = LT.blk[j];
        const int shift = (j % 2 == 0) ? 0 : 2;
        const int* map;
        if ((rc = get_map(h, H, W, shift, &map))) return rc;
        bt.x0 = const_cast<float*>(x);
        bt.xn1 = tp.take((size_t)Ms * L.Cp); bt.qkv = tp.take((size_t)Ms * L.Nqkv); bt.obuf = tp.take((size_t)Ms * L.Ko);
        const bool fmlp = mlp_train_fused(L);
        bt.x1 = tp.take((size_t)M * L.Cp);
        bt.xn2 = bt.hpre = bt.hact = nullptr;
        if (!fmlp) { bt.xn2 = tp.take((size_t)M * L.Cp); bt.hpre = tp.take((size_t)M * L.hiddenP); bt.hact = tp.take((size_t)M * L.hiddenP); }
        bt.x2 = tp.take((size_t)M * L.Cp);
        if (!bt.x2) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
        // Round 4: at the memory-bound widths (C <= 96) LayerNorm + QKV projection + window attention + output projection are ONE launch of the
        // inference path's fused kernel, which also writes the tape entries (xn1, q | k | v, attention output) in the layouts the backward reads
        // (fused_attn.h, TAPE): 1.73 GB -> 0.93 GB of HBM traffic per C = 45 block at 36 clips.  ESCX_TRAIN_ATTN_FUSED=0: the four launches.
        static const bool attn_fused_ok = [] { const char* e = getenv("ESCX_TRAIN_ATTN_FUSED"); return !(e && e[0] == '0'); }();
        int afrc = -1;
        static const int attn_fused_max = [] { const char* e = ESCX_TUNE_ENV("ESCX_TRAIN_ATTN_FUSED_MAXCP"); return e && e[0] ? atoi(e) : 96; }();
        // the TAPE instantiation zeroes 16 pad columns of the q|k|v and attention-output rows; a wider pad (no ESC geometry has one) keeps the unfused launches, whose
        // GEMM epilogues write whole rows - the backward contracts over the full Nqkv / Ko width
        const bool tape_pad_ok = L.Nqkv - 3 * L.nH * L.hdp <= 16 && L.Ko - L.nH * L.hdp <= 16;
        if (attn_fused_ok && tape_pad_ok && L.attn_mode >= 0 && L.Cp <= attn_fused_max) {
            const AttnTape tape{bt.xn1, bt.qkv, bt.obuf, L.Nqkv, L.Ko, L.hdp, L.nH};
            const int tmw = attn_windows_per_wave(L.Cp);
            const long long wg4 = (Ms / 16 / tmw + 3) / 4;
            const int nw = wg4 >= 2048 ? 8 : 4;
            int gs = 1;
            const size_t n_recs = h->prof_recs.size();
            PROF("T.attn_fused" + tg, 8 * dMs * dC * dC + 4 * dMs * 16 * dC, (dM * 2 + dMs * 6) * dC * f4,
                 afrc = attn_fused(x, bt.x1, L.Cp, L.C, L.attn_mode, L.n_groups, bw.ln1_g, bw.ln1_b, bw.waf, bw.baf, bw.bias_tab_f, bw.bproj, map, slots, tokens,
                                   Ms / 16, Hp / 4, Wp / 4, shift > 0, 1.0f / std::sqrt((float)L.hd), nw, &gs, nullptr, M, st, nullptr, &tape));
            if (afrc != 0 && h->prof_recs.size() > n_recs) h->prof_recs.pop_back();
        }
        if (afrc != 0) {
        PROF("T.ln1_gather" + tg, 0, (dM + dMs) * dC * f4, ln_rows(1, x, bt.xn1, bw.ln1_g, bw.ln1_b, map, slots, tokens, Ms, L.C, L.Cp, st));
        PROF("T.gemm_qkv" + tg, 2 * dMs * dC * 3 * dC, dMs * 4 * dC * f4,
             gemm_rows(bt.xn1, L.Cp, Ms, bw.wqkv, L.Nqkv, L.Cp, EpiQkv{bt.qkv, L.Nqkv, bw.bqkv, L.nH * L.hdp, 1.0f / std::sqrt((float)L.hd)}, st));
        int arc = 0;
        PROF("T.window_attn" + tg, 4 * dMs * 16 * dC, dMs * 4 * dC * f4,
             arc = window_attention(bt.qkv, bw.bias_tab, bt.obuf, Ms / 16, L.nH, L.hdp, L.Nqkv, L.Ko, Hp / 4, Wp / 4, shift > 0, st));
        if (arc) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "head_dim %d unsupported by the attention kernel", L.hd);
        PROF("T.gemm_proj" + tg, 2 * dMs * dC * dC, (dMs * dC + 2 * dM * dC) * f4,
             gemm_rows(bt.obuf, L.Ko, Ms, bw.wproj, L.Cp, L.Ko, EpiProjScatter{bt.x1, x, bw.bproj, map, slots, tokens, L.Cp}, st));
        }
        if (fmlp) {       // LN2 + fc1 + GELU + fc2 + residual in one kernel: x1 -> x2 (x1 itself is what the backward recomputes from)
            int hs = 1, frc = 0;
            if (mlp_train_x3() && bw.x3w_train && !h->train_x3_stale)
                PROF("T.mlp_x3" + tg, 4 * dM * dC * L.hidden, 2 * dM * dC * f4,
                     frc = mlp_x3(bt.x1, M, L.C, L.Cp, bw.ln2_g, bw.ln2_b, bw.b1, bw.b2, bw.x3w_train, L.hiddenP, (M + 15) / 16 >= 8 * 512 ? 8 : 4, &hs, nullptr, st, nullptr, 3, bt.x2));
            else
            PROF("T.mlp_fused" + tg, 4 * dM * dC * L.hidden, 2 * dM * dC * f4,
                 frc = mlp_fused(bt.x1, M, L.C, L.Cp, bw.ln2_g, bw.ln2_b, bw.w1f, bw.b1, bw.w2f, bw.b2, bw.wcf, L.hiddenP, 3, &hs, nullptr, st, bt.x2));
            if (frc) ESCX_FAIL(ESCX_ERR_STATE, "fused MLP not instantiated for Cp = %d", L.Cp);
            x = bt.x2;
            continue;
        }
        PROF("T.ln2" + tg, 0, 2 * dM * dC * f4, ln_rows(0, bt.x1, bt.xn2, bw.ln2_g, bw.ln2_b, nullptr, tokens, tokens, M, L.C, L.Cp, st));
        PROF("T.gemm_fc1_gelu" + tg, 2 * dM * dC * L.hidden, dM * (dC + 2 * L.hidden) * f4,
             gemm_rows(bt.xn2, L.Cp, M, bw.w1, L.hiddenP, L.Cp, EpiGeluDual{bt.hpre, bt.hact, L.hiddenP, bw.b1}, st));
        PROF("T.gemm_fc2_res" + tg, 2 * dM * dC * L.hidden, dM * (2 * dC + L.hidden) * f4,
             gemm_rows(bt.hact, L.hiddenP, M, bw.w2, L.Cp, L.hiddenP, EpiResidual{bt.x2, L.Cp, bw.b2, bt.x1}, st));
        x = bt.x2;
    }
    if (L.scale == 1) {
        const int H2 = (H + 1) / 2, M2 = B * H2 * W;
        const int* map;
        if ((rc = get_map(h, H, W, -1, &map))) return rc;
        LT.sub_xn = tp.take((size_t)M2 * 2 * L.Cp); LT.y = tp.take((size_t)M2 * L.CoutP);
        if (!LT.y) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
        PROF("T.merge_ln", 0, 2.0 * M * L.C * 4, ln_rows(2, x, LT.sub_xn, L.sub_g, L.sub_b, map, H2 * W, tokens, M2, L.C, L.Cp, st));
        PROF("T.merge_gemm", 2.0 * M2 * 2 * L.C * L.Cout, (double)M2 * (2 * L.C + L.Cout) * 4,
             gemm_store(LT.sub_xn, 2 * L.Cp, M2, L.sub_w, L.CoutP, 2 * L.Cp, LT.y, L.CoutP, nullptr, st));
        LT.Hout = H2;
    } else if (L.scale == 2) {
        LT.sub_xn = tp.take((size_t)M * L.Cp); LT.y = tp.take((size_t)2 * M * L.CoutP);
        if (!LT.y) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
        PROF("T.split_ln", 0, 2.0 * M * L.C * 4, ln_rows(0, x, LT.sub_xn, L.sub_g, L.sub_b, nullptr, tokens, tokens, M, L.C, L.Cp, st));
        PROF("T.split_gemm", 2.0 * M * L.C * 2 * L.Cout, (double)M * (L.C + 2 * L.Cout) * 4,
             gemm_split(LT.sub_xn, L.Cp, M, L.sub_w, 2 * L.CoutP, L.Cp, LT.y, H, W, L.CoutP, st));
        LT.Hout = 2 * H;
    } else {
        LT.y = const_cast<float*>(x); LT.Hout = H;
    }
    return launch_ok(L.prefix.c_str());
}

// one cross-scale VQ step in training mode (csrvq.py:23-48).  dec == nullptr for stream 0 (enc - 0.0).
int quant_fwd(escx_handle_s* h, TrainTape& T, int sid, const float* enc, const float* dec, bool transmit, float* zpart, hipStream_t st) {
    const escx_config& c = h->cfg;
    const Quant& q = h->quants[sid];
    const Shapes& s = T.shp;
    Arena& tp = h->tape;
    QuantTape& Q = T.q[sid];
    const int B = s.B, W = s.W, Tq = s.Tq, M = B * Tq, G = c.group_size;
    const long long bstride = (long long)c.max_streams * G * Tq;
    long long* codes = T.codes + (long long)sid * G * Tq;
    Q.transmit = transmit; Q.enc = enc; Q.dec = dec;
    Q.ze = tp.take((size_t)M * q.Nz); Q.zup = tp.take((size_t)M * q.Nz);
    if (!Q.zup) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
    const int splits = pvq_down_splits(M, q.Kq, q.Cp);
    const double vec = (double)c.overlap * q.Hq * q.C;
    PROF("T.pvq_down", 2.0 * M * vec * q.d, (double)M * vec * (dec ? 2 : 1) * 4,
         gemm_pvq_down(enc, dec, B, q.Hq, W, q.Cp, c.overlap, q.wd, q.Nz, q.Kq, zpart, splits, st));
    reduce_partials(zpart, splits, (long long)M * q.Nz, Q.ze, 0, st);
    int src = 0;
    PROF("T.pvq_search", 2.0 * M * G * c.codebook_size * q.d, (double)G * c.codebook_size * q.d * 4,
         src = pvq_search(Q.ze, 1, M, q.Nz, q.cbn, q.c2, q.cbraw, G, c.codebook_size, q.d, q.dt, Tq, codes, bstride, nullptr, 0.f, c.l2norm, st));
    if (src) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "codebook_dim %d unsupported by the search kernel", q.d);
    if (!transmit) { Q.out = const_cast<float*>(dec); return launch_ok("quant_fwd"); }      // residual_q *= 0 (csrvq.py:42-44): the refined map IS dec
    float* terms = T.terms + (size_t)sid * G * M;
    hipLaunchKernelGGL(pvq_train_fwd_kernel, dim3(blocks_for((long long)M * G)), dim3(256), 0, st, Q.ze, codes, bstride, q.cbraw, Q.zup, terms, M, G,
                       c.codebook_size, q.d, q.dt, q.Nz, Tq, 1.0f / ((float)Tq * q.d * G), T.freeze);
    Q.out = tp.take((size_t)B * q.Hq * W * q.Cp);
    if (!Q.out) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
    PROF("T.pvq_up", 2.0 * M * vec * q.d, (double)M * vec * 2 * 4,
         gemm_any(PlainA{Q.zup, q.Nz, M}, q.wup, M, q.Kq, q.Kup,
                  EpiPvqAdd{Q.out, dec, q.Hq, W, q.Cp, Tq, c.overlap, FastDiv(Tq), FastDiv(q.Cp), FastDiv(q.Hq)}, st));
    return launch_ok("quant_fwd");
}

int refresh_from_flat(escx_handle_s* h, const float* flat, hipStream_t st) {
    int rc = build_gather_map(h);
    if (rc) return rc;
    const long long n = (long long)(h->wts.cap / sizeof(float));
    hipLaunchKernelGGL(gather_params_kernel, dim3(blocks_for(n)), dim3(256), 0, st, flat, h->gmap, reinterpret_cast<float*>(h->wts.base), n);
    const escx_config& c = h->cfg;
    for (const Quant& q : h->quants) {
        const int rows = c.group_size * c.codebook_size;
        hipLaunchKernelGGL(codebook_normalize_kernel, dim3(blocks_for(rows)), dim3(256), 0, st, q.cbraw, q.cbn, q.c2, rows, q.d, q.dt, c.l2norm);
    }
    // the folded 7x7 de-embedding (inference path, and the training forward below) from the refreshed convolution weights: fp64 on the device, the
    // host fold's summation order - bit-identical to escx_finalize_params, so the model decodes correctly after device-side optimiser steps
    if (c.in_dim * h->Q <= 16 && h->dcv_w && h->dcc_w && h->dch_w) {
        const long long n = (long long)16 * c.in_dim * h->Q * 49 * h->C0 + 16 * c.in_dim * h->Q;
        hipLaunchKernelGGL(deembed_compose_kernel, dim3(blocks_for(n)), dim3(256), 0, st, h->dc1_w, h->dc1_b, h->dc2_w, h->dc2_b, h->dcv_w, h->dcv_b,
                           h->dcc_w, h->dcc_b, h->dch_w, h->C0, h->C0p, c.patch_f, c.patch_t, c.in_dim);
        h->composed_stale = false;
    } else {
        h->composed_stale = true;       // no folded form for this geometry: the inference path needs escx_load_flat_params(full = 1)
    }
    h->train_x3_stale = true;
    h->pvq_tab_stale = true;            // derived inference state (de-quantisation tables, split MLP weight images): rebuilt by the next inference entry
    return launch_ok("refresh_from_flat");
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" int escx_flat_param_count(escx_handle h) { return h ? (int)h->flat_keys.size() : 0; }
extern "C" const char* escx_flat_param_key(escx_handle h, int i) { return (h && i >= 0 && i < (int)h->flat_keys.size()) ? h->flat_keys[i].c_str() : nullptr; }
extern "C" int64_t escx_flat_param_offset(escx_handle h, int i) { return (h && i >= 0 && i < (int)h->flat_off.size()) ? (int64_t)h->flat_off[i] : -1; }
extern "C" int64_t escx_flat_param_numel(escx_handle h, int i) { return (h && i >= 0 && i < (int)h->flat_numel.size()) ? (int64_t)h->flat_numel[i] : -1; }
extern "C" int64_t escx_flat_param_total(escx_handle h) { return h ? (int64_t)h->flat_total : 0; }

extern "C" int escx_load_flat_params(escx_handle h, const float* flat_dev, int full, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!flat_dev) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    if (!full) return refresh_from_flat(h, flat_dev, (hipStream_t)stream);
    // full: host round trip, so that the fp64-folded layouts of the inference path are rebuilt too
    ESCX_HIP(hipStreamSynchronize((hipStream_t)stream));
    std::vector<float> host(h->flat_total);
    ESCX_HIP(hipMemcpy(host.data(), flat_dev, h->flat_total * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h->flat_keys.size(); ++i) {
        Param& p = h->params[h->flat_keys[i]];
        std::copy(host.begin() + h->flat_off[i], host.begin() + h->flat_off[i] + h->flat_numel[i], p.data.begin());
    }
    rc = escx_finalize_params(h);
    if (!rc) h->composed_stale = false;
    return rc;
}

extern "C" int64_t escx_train_tape_bytes(escx_handle h) {
    if (!h) return 0;
    // everything the handle holds for activation tapes: the one-part arena AND the per-part arenas when both forms have been used
    const TrainRoot* r = static_cast<TrainRoot*>(h->train_state);
    int64_t t = (int64_t)h->tape.cap;
    if (r) for (int p = 0; p < TrainRoot::MAXP; ++p) t += (int64_t)r->arena[p].cap;
    return t;
}
extern "C" int64_t escx_train_tape_generation(escx_handle h) { return (h && h->train_state) ? (int64_t)static_cast<TrainRoot*>(h->train_state)->generation : 0; }

namespace {
// one pass over B clips on stream st, tape = tape_of(h) in the arena h->tape (the caller points both at the part it wants)
// feat != nullptr: the spectrum is given ((B, T, in_dim, F) frame-major, forward(x, x_feat=...) of codecs.py:33-34) and the STFT is skipped; L = hop * (T - 1) then
int train_forward_impl(escx_handle_s* h, const float* wave, const float* feat, int B, int L, int S, int freeze, int64_t* codes_out, float* wave_out, float* raw_feat,
                       float* recon_feat, float* cm_loss, float* cb_loss, hipStream_t st) {
    int rc = 0;
    const escx_config& c = h->cfg;
    TrainTape& T = *tape_of(h);
    T.valid = false;
    ++T.generation;
    Shapes s;
    if ((rc = make_shapes(h, B, 1 + L / c.hop_length, &s))) return rc;
    const size_t need = tape_bytes(h, s);
    if (h->tape.cap < need) {
        ESCX_HIP(hipDeviceSynchronize());
        if (h->tape.base) ESCX_HIP(hipFree(h->tape.base));
        h->tape = Arena();
        ESCX_HIP(hipMalloc((void**)&h->tape.base, need));
        h->tape.cap = need;
    }
    Arena& tp = h->tape;
    tp.used = 0;
    T.B = B; T.L = L; T.S = freeze ? c.max_streams : S; T.freeze = freeze ? 1 : 0; T.shp = s;      // codecs.py:65: frozen codebooks use every stream
    const int n = h->n, G = c.group_size, Smax = c.max_streams;
    const int T2 = c.patch_t * s.W, out_len = c.hop_length * (T2 - 1);

    // STFT -> patch embedding (base.py:29-37, scale.py:42-50)
    T.spec = tp.take((size_t)B * s.T * c.in_dim * h->Fp);
    const size_t tok0 = (size_t)B * s.H0 * s.W * h->C0p;
    T.pe_pre = tp.take(tok0); T.tok0 = tp.take(tok0);
    T.terms = tp.take((size_t)Smax * G * B * s.Tq);
    T.codes = reinterpret_cast<long long*>(tp.take((size_t)B * Smax * G * s.Tq * 2));
    if (!T.codes) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
    if (feat) pad_rows(feat, T.spec, (long long)B * s.T * c.in_dim, h->F, h->Fp, st);
    else
    PROF("T.stft", 2.0 * B * s.T * c.win_length * 2 * h->F, ((double)B * L + (double)B * s.T * 2 * h->F) * 4,
         gemm_frames(wave, B, L, s.T, c.hop_length, h->left - h->n_fft / 2, h->dft_w, c.in_dim * h->Fp, h->winP, T.spec, st));
    if (raw_feat) unpad_rows(T.spec, raw_feat, (long long)B * s.T * c.in_dim, h->F, h->Fp, st);
    PROF("T.patch_embed", 2.0 * tok0 * c.in_dim * c.patch_f * c.patch_t, ((double)B * s.T * 2 * h->F + 2.0 * tok0) * 4,
         gemm_patch(T.spec, B, s.T, c.in_dim, h->Fp, s.H0, s.W, c.patch_f, c.patch_t, h->pe_w, h->C0p, h->Kpe, T.pe_pre, h->pe_b, st));
    ln_rows(0, T.pe_pre, T.tok0, h->pe_g, h->pe_beta, nullptr, s.H0 * s.W, s.H0 * s.W, B * s.H0 * s.W, h->C0, h->C0p, st);

    // encoder (base.py:143-158)
    T.layers.assign(2 * n, LayerTape());
    T.enc_hs.assign(n, nullptr);
    if ((rc = layer_fwd(h, h->layers[0], T.layers[0], T.tok0, B, s.H0, s.W, st))) return rc;
    T.enc_hs[0] = T.layers[0].y;
    for (int i = 0; i + 1 < n; ++i) {
        if ((rc = layer_fwd(h, h->layers[1 + i], T.layers[1 + i], T.enc_hs[i], B, s.encH[i], s.W, st))) return rc;
        T.enc_hs[i + 1] = T.layers[1 + i].y;
    }

    // cross-scale VQ decoder in training mode (csrvq.py:97-129): every stream is quantised (codes of all max_streams are returned),
    // streams >= num_streams are masked out of the reconstruction and of the losses
    T.q.assign(Smax, QuantTape());
    size_t zp = 0;
    for (const Quant& q : h->quants) zp = std::max(zp, (size_t)pvq_down_splits(B * s.Tq, q.Kq, q.Cp) * B * s.Tq * q.Nz);
    float* zpart = tp.take(zp);              // split-K partials of the down-projections (dead once z_e is reduced; reused by every stream)
    if (!zpart) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
    if ((rc = quant_fwd(h, T, 0, T.enc_hs[n - 1], nullptr, true, zpart, st))) return rc;
    const float* dec = T.q[0].out;
    int H = s.encH[n - 1];
    for (int i = 0; i + 1 < n; ++i) {
        const bool transmit = i < T.S - 1;
        if ((rc = quant_fwd(h, T, i + 1, T.enc_hs[n - 1 - i], dec, transmit, zpart, st))) return rc;
        if ((rc = layer_fwd(h, h->layers[n + i], T.layers[n + i], T.q[i + 1].out, B, H, s.W, st))) return rc;
        dec = T.layers[n + i].y; H = T.layers[n + i].Hout;
    }
    if ((rc = layer_fwd(h, h->layers[2 * n - 1], T.layers[2 * n - 1], dec, B, H, s.W, st))) return rc;
    T.post = T.layers[2 * n - 1].y;

    // De-embedding (scale.py:73-81).  Round 4: the FOLDED 7x7 form of the inference path (composed on the device from the current weights in
    // refresh_from_flat): 11x fewer FLOPs than the two convolutions, and the 270-channel fine map (0.75 GB at 36 clips) never exists - the backward
    // gets conv3x3's weight gradient from the product it already forms for conv5x5's (deembed_x_from_r_kernel).  The two convolutions of the
    // reference remain for geometries without a folded form and as the A/B baseline (ESCX_TRAIN_DEEMBED_COMPOSED=0).
    const int F2 = c.patch_f * s.H0;
    T.composed = train_deembed_composed(h);
    if (T.composed && h->composed_stale) ESCX_FAIL(ESCX_ERR_STATE, "folded de-embedding is stale in a training forward (refresh_from_flat did not run)");
    T.deemb = T.composed ? nullptr : tp.take((size_t)B * T2 * F2 * h->C0p);
    T.rspec = tp.take((size_t)B * T2 * c.in_dim * h->Fp);
    if (!T.rspec || (!T.composed && !T.deemb)) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
    ESCX_HIP(hipMemsetAsync(T.rspec, 0, (size_t)B * T2 * c.in_dim * h->Fp * sizeof(float), st));       // Fp - F pad columns stay zero
    if (T.composed) {
        const double tk = (double)B * s.H0 * s.W;
        int hrc = -1;
        PROF("T.deembed_composed7x7", 2.0 * tk * 49 * h->C0 * c.in_dim * h->Q, (tk * h->C0 + tk * c.in_dim * h->Q) * 4,
             hrc = deembed7_fused(T.post, B, s.H0, s.W, h->C0p, h->dch_w, h->dcc_b, T.rspec, c.patch_f, c.patch_t, c.in_dim, h->Fp, st));
        if (hrc != 0)
            PROF("T.deembed_composed7x7", 2.0 * tk * 49 * h->C0 * c.in_dim * h->Q, (tk * h->C0 + tk * c.in_dim * h->Q) * 4,
                 gemm_deembed_composed(T.post, B, s.H0, s.W, h->C0p, h->dcc_w, T.rspec, h->dcc_b, c.patch_f, c.patch_t, c.in_dim, h->Fp, st));
        int brc = 0;
        PROF("T.deembed_border", 0, 0,
             brc = deembed_border(T.post, h->dcv_w, h->dcv_b, T.rspec, B, s.H0, s.W, h->C0, h->C0p, c.patch_f, c.patch_t, c.in_dim, h->Fp, st));
        if (brc) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "patch size unsupported by the de-embedding border kernel");
    } else {
        const double toks = (double)B * s.H0 * s.W, pix = toks * h->Q;
        PROF("T.deembed_conv5x5", 2.0 * toks * 25 * h->C0 * h->C0 * h->Q, (toks * h->C0 + pix * h->C0) * 4,
             gemm_conv_deembed1(T.post, B, s.H0, s.W, h->C0p, h->dc1_w, h->Q * h->C0p, T.deemb, h->dc1_b, c.patch_f, c.patch_t, st));
        PROF("T.deembed_conv3x3", 2.0 * pix * 9 * h->C0 * c.in_dim, (pix * h->C0 + pix * c.in_dim) * 4,
             gemm_conv_spec(T.deemb, B, T2, F2, h->C0p, h->dc2_w, T.rspec, h->dc2_b, h->Fp, c.in_dim, st));
    }
    if (recon_feat) unpad_rows(T.rspec, recon_feat, (long long)B * T2 * c.in_dim, h->F, h->Fp, st);
    {   // inverse STFT (base.py:39-47)
        Scratch sc(&tp);
        float* frames = sc.take((size_t)B * T2 * h->winP);
        if (!frames) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small");
        PROF("T.istft", 2.0 * B * T2 * c.win_length * 2 * h->F, (double)B * T2 * (2 * h->F + c.win_length) * 4,
             gemm_store(T.rspec, c.in_dim * h->Fp, B * T2, h->idft_w, h->winP, c.in_dim * h->Fp, frames, h->winP, nullptr, st));
        istft_ola(frames, h->win2, wave_out, B, T2, h->winP, c.win_length, c.hop_length, h->left, h->n_fft / 2, out_len, st);
    }
    // per-clip VQ losses (codebook.py:67-69; quantization.py:57-59,71-72; csrvq.py:42-44): sum over the transmitted streams
    if (cm_loss) loss_reduce(T.terms, T.S, G, B * s.Tq, s.Tq, cm_loss, st);
    if (cb_loss) loss_reduce(T.terms, T.S, G, B * s.Tq, s.Tq, cb_loss, st);
    ESCX_HIP(hipMemcpyAsync(codes_out, T.codes, (size_t)B * Smax * G * s.Tq * sizeof(long long), hipMemcpyDeviceToDevice, st));
    T.fwd_mark = tp.used;
    T.valid = true;
    return launch_ok("train_forward");
}

int train_parts_for(escx_handle_s* h, int B) {
    const char* e = getenv("ESCX_TRAIN_PARTS");                 // read per call: tests switch it
    const int want = e ? atoi(e) : 2;
    const char* m = getenv("ESCX_TRAIN_PARTS_MIN_BATCH");
    const int min_b = m ? atoi(m) : 8;
    if (want < 2 || B < std::max(2, min_b) || h->prof) return 1;
    return std::min(std::min(want, TrainRoot::MAXP), B);
}
}  // namespace

static int train_forward_entry(escx_handle h, const float* flat_dev, const float* wave, const float* feat, int B, int L, int S, int freeze, int64_t* codes_out,
                               float* wave_out, float* raw_feat, float* recon_feat, float* cm_loss, float* cb_loss, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if ((!wave && !feat) || !codes_out || !wave_out) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    const escx_config& c = h->cfg;
    if (B < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "batch must be positive");
    if (S < 1 || S > c.max_streams) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "num_streams=%d outside [1, %d]", S, c.max_streams);
    if (wave && L <= h->n_fft / 2) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_samples=%d too short for reflect padding of %d", L, h->n_fft / 2);
    if (h->ws != 4) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "the training step is implemented for window_size=4 (got %d): inference only for other window sizes", h->ws);
    hipStream_t st = (hipStream_t)stream;
    if (flat_dev && (rc = refresh_from_flat(h, flat_dev, st))) return rc;
    if ((rc = build_gather_map(h))) return rc;
    if ((rc = pack_train_mlp_images(h, st))) return rc;         // before the batch parts fork
    TrainRoot& R = *root_of(h);
    R.valid = false;
    ++R.generation;
    R.parts = train_parts_for(h, B);
    {   // only the arenas of the form in use stay allocated (a run that alternates between one part and several - profiling, mixed batch
        // sizes - used to hold both: ~35 GB + 2 x 17 GB at 36 clips)
        auto drop = [&](Arena& a) -> int { if (a.base) { ESCX_HIP(hipDeviceSynchronize()); ESCX_HIP(hipFree(a.base)); a = Arena(); } return 0; };
        if (R.parts == 1) { for (int p = 0; p < TrainRoot::MAXP; ++p) if ((rc = drop(R.arena[p]))) return rc; }
        else if ((rc = drop(h->tape))) return rc;
    }
    if (R.parts == 1) {
        R.cur = &R.single;
        rc = train_forward_impl(h, wave, feat, B, L, S, freeze, codes_out, wave_out, raw_feat, recon_feat, cm_loss, cb_loss, st);
        R.valid = rc == 0;
        return rc;
    }
    if (!R.ev_fork) ESCX_HIP(hipEventCreateWithFlags(&R.ev_fork, hipEventDisableTiming));
    ESCX_HIP(hipEventRecord(R.ev_fork, st));
    for (int p = 1; p < R.parts; ++p) {
        if (!R.aux[p]) {
            ESCX_HIP(hipStreamCreateWithFlags(&R.aux[p], hipStreamNonBlocking));
            ESCX_HIP(hipEventCreateWithFlags(&R.ev_join[p], hipEventDisableTiming));
        }
        ESCX_HIP(hipStreamWaitEvent(R.aux[p], R.ev_fork, 0));
    }
    Shapes s0;
    if ((rc = make_shapes(h, 1, 1 + L / c.hop_length, &s0))) return rc;
    const size_t per_codes = (size_t)c.max_streams * c.group_size * s0.Tq;
    const size_t per_wave_out = (size_t)c.hop_length * (c.patch_t * s0.W - 1);
    const size_t per_raw = (size_t)s0.T * c.in_dim * h->F, per_recon = (size_t)c.patch_t * s0.W * c.in_dim * h->F;
    for (int p = 0; p <= R.parts; ++p) R.first[p] = (int)((long long)B * p / R.parts);
    for (int p = 0; p < R.parts; ++p) {
        const int b0 = R.first[p], nb = R.first[p + 1] - b0;
        std::swap(h->tape, R.arena[p]);
        R.cur = &R.part[p];
        rc = train_forward_impl(h, wave ? wave + (size_t)b0 * L : nullptr, feat ? feat + b0 * per_raw : nullptr, nb, L, S, freeze, codes_out + b0 * per_codes, wave_out + b0 * per_wave_out,
                                raw_feat ? raw_feat + b0 * per_raw : nullptr, recon_feat ? recon_feat + b0 * per_recon : nullptr,
                                cm_loss ? cm_loss + b0 : nullptr, cb_loss ? cb_loss + b0 : nullptr, p ? R.aux[p] : st);
        std::swap(h->tape, R.arena[p]);
        R.cur = &R.single;
        if (rc) break;                  // join the aux streams below even on an error, they may still run on caller tensors
    }
    for (int p = 1; p < R.parts; ++p) {
        ESCX_HIP(hipEventRecord(R.ev_join[p], R.aux[p]));
        ESCX_HIP(hipStreamWaitEvent(st, R.ev_join[p], 0));
    }
    R.valid = rc == 0;
    return rc;
}

extern "C" int escx_train_forward(escx_handle h, const float* flat_dev, const float* wave, int B, int L, int S, int freeze, int64_t* codes_out,
                                  float* wave_out, float* raw_feat, float* recon_feat, float* cm_loss, float* cb_loss, void* stream) {
    if (!wave) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    return train_forward_entry(h, flat_dev, wave, nullptr, B, L, S, freeze, codes_out, wave_out, raw_feat, recon_feat, cm_loss, cb_loss, stream);
}

// ESC.forward in training mode with a precomputed spectrum (forward(x, x_feat=...), codecs.py:33-34): feat_dev is (B, T, in_dim, F) frame-major like escx_forward_feat's
extern "C" int escx_train_forward_feat(escx_handle h, const float* flat_dev, const float* feat, int B, int n_frames, int S, int freeze, int64_t* codes_out,
                                       float* wave_out, float* recon_feat, float* cm_loss, float* cb_loss, void* stream) {
    if (!h) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null handle");
    if (!feat) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null pointer");
    if (n_frames < h->cfg.patch_t) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_frames=%d shorter than one patch", n_frames);
    return train_forward_entry(h, flat_dev, nullptr, feat, B, h->cfg.hop_length * (n_frames - 1), S, freeze, codes_out, wave_out, nullptr, recon_feat, cm_loss, cb_loss, stream);
}

namespace {

// backward of one TransformerLayer.  gy: gradient of the layer output (rows of LT.y); returns the gradient of the layer INPUT in *gx
// (a scratch buffer owned by `sc`, valid until the caller's Scratch dies).
int layer_bwd(escx_handle_s* h, const Layer& L, const LayerTape& LT, const float* gy, int B, int W, float* gflat, Scratch& sc, float** gx, hipStream_t st) {
    const int H = LT.H, tokens = H * W, M = B * tokens;
    const int Hp = rup(H, 4), Wp = rup(W, 4), slots = Hp * Wp, Ms = B * slots;
    float* part = sc.take(DW_PART_FLOATS);
    float* lnpart = sc.take(LN_PART_FLOATS);
    float* attpart = sc.take(ATT_PART_FLOATS);
    float* dcur = sc.take((size_t)M * L.Cp);                 // gradient w.r.t. the current block output
    float* dx1 = sc.take((size_t)M * L.Cp);
    float* dx1s = sc.take((size_t)Ms * L.Cp);                // the same gradient in window-slot order (pad slots zero)
    float* dxn = sc.take((size_t)std::max(M, Ms) * L.Cp);
    float* dhpre = sc.take((size_t)M * L.hiddenP);
    float* dqkv = sc.take((size_t)Ms * L.Nqkv);
    float* dobuf = sc.take((size_t)Ms * L.Ko);
    float* dbias = sc.take((size_t)L.nH * 256);
    float* dprev = sc.take((size_t)M * L.Cp);
    if (!dprev) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
    int rc;
    const std::string tg = h->prof ? "[C=" + std::to_string(L.C) + "]" : std::string();
    const float* x_last = LT.blk.back().x2;
    const float* dlast;                                     // gradient w.r.t. the last block's output
    if (L.scale == 1) {
        const int H2 = (H + 1) / 2, M2 = B * H2 * W;
        const int* map;
        if ((rc = get_map(h, H, W, -1, &map))) return rc;
        float* dsub = sc.take((size_t)M2 * 2 * L.Cp);
        if (!dsub) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        if ((rc = dw_rows(h, gy, L.CoutP, LT.sub_xn, 2 * L.Cp, M2, L.CoutP, 2 * L.Cp, G(h, L.sub_w), nullptr, part, st))) return rc;
        gemm_rows(gy, L.CoutP, M2, L.sub_wT, 2 * L.Cp, L.CoutP, EpiStore{dsub, 2 * L.Cp, nullptr}, st);
        ln_bwd(2, x_last, dsub, L.sub_g, map, nullptr, dcur, G(h, L.sub_g), G(h, L.sub_b), H2 * W, tokens, 0, M2, L.C, L.Cp, lnpart, st);
        dlast = dcur;
    } else if (L.scale == 2) {
        float* dsub = sc.take((size_t)M * L.Cp);
        if (!dsub) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        SplitGatherA ga{gy, H, W, L.CoutP, M, FastDiv(H * W), FastDiv(W), FastDiv(L.CoutP)};
        if ((rc = dw_launch(h, ga, PlainA{LT.sub_xn, L.Cp, M}, M, 2 * L.CoutP, L.Cp, G(h, L.sub_w), nullptr, part, st))) return rc;
        gemm_any(ga, L.sub_wT, M, L.Cp, 2 * L.CoutP, EpiStore{dsub, L.Cp, nullptr}, st, 16);
        ln_bwd(0, x_last, dsub, L.sub_g, nullptr, nullptr, dcur, G(h, L.sub_g), G(h, L.sub_b), tokens, tokens, tokens, M, L.C, L.Cp, lnpart, st);
        dlast = dcur;
    } else {
        dlast = gy;
    }
    const float* dy = dlast;
    for (int j = (int)L.blocks.size() - 1; j >= 0; --j) {
        const BlockW& bw = L.blocks[j];
        const BlockTape& bt = LT.blk[j];
        const int shift = (j % 2 == 0) ? 0 : 2;
        const int *map, *inv;
        if ((rc = get_map(h, H, W, shift, &map))) return rc;
        if ((rc = get_map(h, H, W, 10 + shift, &inv))) return rc;
        // ---- MLP: x2 = x1 + W2 gelu(W1 LN2(x1) + b1) + b2 ----
        const bool fmlp = mlp_train_fused(L);
        static const bool ln_fused = [] { const char* e = ESCX_TUNE_ENV("ESCX_LN_FUSED"); return !(e && e[0] == '0'); }();
        if (fmlp) {
            if (slots != tokens) ESCX_HIP(hipMemsetAsync(dx1s, 0, (size_t)Ms * L.Cp * sizeof(float), st));      // pad slots carry no gradient
            PROF("B.mlp_fused" + tg, 10.0 * M * L.C * L.hidden, (3.0 * M + Ms) * L.Cp * 4,
                 rc = mlp_bwd_fused(h, L, bw, bt.x1, dy, dx1, dx1s, inv, tokens, slots, M, part, dxn, dhpre, lnpart, st));
            if (rc) return rc;
        } else {
        PROF("B.dw_fc2" + tg, 2.0 * M * L.C * L.hidden, (double)M * (L.Cp + L.hiddenP) * 4,
             rc = dw_rows(h, dy, L.Cp, bt.hact, L.hiddenP, M, L.Cp, L.hiddenP, G(h, bw.w2), G(h, bw.b2), part, st));
        if (rc) return rc;
        PROF("B.dx_fc2" + tg, 2.0 * M * L.C * L.hidden, (double)M * (L.Cp + 2 * L.hiddenP) * 4,
             gemm_rows(dy, L.Cp, M, bw.w2T, L.hiddenP, L.Cp, EpiGeluBwd{dhpre, L.hiddenP, bt.hpre}, st));
        PROF("B.dw_fc1" + tg, 2.0 * M * L.C * L.hidden, (double)M * (L.Cp + L.hiddenP) * 4,
             rc = dw_rows(h, dhpre, L.hiddenP, bt.xn2, L.Cp, M, L.hiddenP, L.Cp, G(h, bw.w1), G(h, bw.b1), part, st));
        if (rc) return rc;
        if (slots != tokens) ESCX_HIP(hipMemsetAsync(dx1s, 0, (size_t)Ms * L.Cp * sizeof(float), st));      // pad slots carry no gradient
        if (ln_fused && ln_rows_fusable(L.Cp)) {       // LN2's backward rides in the epilogue of the GEMM that produces its upstream gradient
            PROF("B.dx_fc1+ln2" + tg, 2.0 * M * L.C * L.hidden, ((double)M * (L.hiddenP + 3 * L.Cp) + (double)Ms * L.Cp) * 4,
                 gemm_ln_bwd_rows(dhpre, L.hiddenP, M, bw.w1T, L.Cp, L.hiddenP, bt.x1, bw.ln2_g, dy, dx1, dx1s, inv, tokens, slots, L.C, G(h, bw.ln2_g),
                                  G(h, bw.ln2_b), part, st));
        } else {
            PROF("B.dx_fc1" + tg, 2.0 * M * L.C * L.hidden, (double)M * (L.hiddenP + L.Cp) * 4,
                 gemm_rows(dhpre, L.hiddenP, M, bw.w1T, L.Cp, L.hiddenP, EpiStore{dxn, L.Cp, nullptr}, st));
            PROF("B.ln2" + tg, 0, 5.0 * M * L.C * 4,
                 ln_bwd(0, bt.x1, dxn, bw.ln2_g, nullptr, dy, dx1, G(h, bw.ln2_g), G(h, bw.ln2_b), tokens, tokens, tokens, M, L.C, L.Cp, lnpart, st, dx1s, inv, slots));
        }
        }
        // ---- attention: x1 = x0 + scatter(Wp attn(Wqkv gather(LN1(x0)))) ----
        PROF("B.dw_proj" + tg, 2.0 * Ms * L.C * L.C, (double)Ms * (L.Cp + L.Ko) * 4,
             rc = dw_rows(h, dx1s, L.Cp, bt.obuf, L.Ko, Ms, L.Cp, L.Ko, G(h, bw.wproj), G(h, bw.bproj), part, st));
        if (rc) return rc;
        PROF("B.dx_proj" + tg, 2.0 * Ms * L.C * L.C, (double)Ms * (L.Cp + L.Ko) * 4, gemm_rows(dx1s, L.Cp, Ms, bw.wprojT, L.Ko, L.Cp, EpiStore{dobuf, L.Ko, nullptr}, st));
        int arc = 0;
        PROF("B.attn_core" + tg, 10.0 * Ms * 16 * L.C, (double)Ms * (2 * L.Nqkv + L.Ko) * 4,
             arc = attn_bwd(bt.qkv, bw.bias_tab, dobuf, dqkv, dbias, attpart, Ms / 16, L.nH, L.hdp, L.Nqkv, L.Ko, Hp / 4, Wp / 4, shift > 0,
                            1.0f / std::sqrt((float)L.hd), st));
        if (arc) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, arc == -2 ? "batch too large for the attention backward kernel's 32-bit offsets (head_dim %d)" : "head_dim %d unsupported by the attention backward kernel", L.hd);
        if (bw.tab_off >= 0)
            hipLaunchKernelGGL(bias_table_grad_kernel, dim3(blocks_for(49 * L.nH)), dim3(256), 0, st, dbias, gflat + bw.tab_off, L.nH);
        PROF("B.dw_qkv" + tg, 2.0 * Ms * L.C * 3 * L.C, (double)Ms * (L.Nqkv + L.Cp) * 4,
             rc = dw_rows(h, dqkv, L.Nqkv, bt.xn1, L.Cp, Ms, L.Nqkv, L.Cp, G(h, bw.wqkv), G(h, bw.bqkv), part, st));
        if (rc) return rc;
        if (ln_fused && ln_rows_fusable(L.Cp)) {       // LN1's backward in the epilogue of the QKV dX GEMM: its rows are window slots, map = slot -> token
            PROF("B.dx_qkv+ln1" + tg, 2.0 * Ms * L.C * 3 * L.C, ((double)Ms * L.Nqkv + 3.0 * M * L.Cp) * 4,
                 gemm_ln_bwd_rows(dqkv, L.Nqkv, Ms, bw.wqkvT, L.Cp, L.Nqkv, bt.x0, bw.ln1_g, dx1, dprev, nullptr, nullptr, tokens, slots, L.C, G(h, bw.ln1_g),
                                  G(h, bw.ln1_b), part, st, map));
        } else {
            PROF("B.dx_qkv" + tg, 2.0 * Ms * L.C * 3 * L.C, (double)Ms * (L.Nqkv + L.Cp) * 4, gemm_rows(dqkv, L.Nqkv, Ms, bw.wqkvT, L.Cp, L.Nqkv, EpiStore{dxn, L.Cp, nullptr}, st));
            PROF("B.ln1" + tg, 0, 4.0 * M * L.C * 4,
                 ln_bwd(1, bt.x0, dxn, bw.ln1_g, inv, dx1, dprev, G(h, bw.ln1_g), G(h, bw.ln1_b), tokens, tokens, slots, M, L.C, L.Cp, lnpart, st));
        }
        std::swap(dcur, dprev);
        dy = dcur;
    }
    *gx = dcur;
    return launch_ok("layer_bwd");
}

// backward of one transmitted quantiser step: gref = gradient of the refined map (post_fuse output).  Accumulates into d_enc (gradient of the
// encoder map) and, if ddec != nullptr, subtracts from it (pre_fuse: residual = enc - dec).  The +dec path of post_fuse is the caller's.
int quant_bwd(escx_handle_s* h, TrainTape& T, int sid, const float* gref, float* d_enc, float* ddec, const float* dcm, const float* dcb, Scratch& sc,
              hipStream_t st) {
    const escx_config& c = h->cfg;
    const Quant& q = h->quants[sid];
    const QuantTape& Q = T.q[sid];
    const Shapes& s = T.shp;
    const int B = s.B, W = s.W, Tq = s.Tq, M = B * Tq, Gr = c.group_size;
    const long long bstride = (long long)c.max_streams * Gr * Tq;
    const long long* codes = T.codes + (long long)sid * Gr * Tq;
    float* part = sc.take(DW_PART_FLOATS);
    float* dzup = sc.take((size_t)M * q.Nz);
    float* dze = sc.take((size_t)M * q.Nz);
    float* gq = sc.take((size_t)M * q.Nz);
    if (!gq) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
    int rc;
    ResidualGatherA gfr{gref, nullptr, q.Hq, W, q.Cp, Tq, c.overlap, M, FastDiv(Tq), FastDiv(q.Cp), FastDiv(q.Hq)};        // framed view of the map gradient
    // up-projection: out_frames = zup . Wup^T   (Wup packed [Kq][Kup])
    if ((rc = dw_launch(h, gfr, PlainA{Q.zup, q.Nz, M}, M, q.Kq, q.Kup, G(h, q.wup), nullptr, part, st))) return rc;
    gemm_any(gfr, q.wupT, M, q.Kup, q.Kq, EpiStore{dzup, q.Nz, nullptr}, st, pick_bk(q.Cp));
    const float scale = 1.0f / ((float)Tq * q.d * Gr);
    hipLaunchKernelGGL(pvq_train_bwd_kernel, dim3(blocks_for((long long)M * Gr)), dim3(256), 0, st, Q.ze, codes, bstride, q.cbraw, dzup, dcm, dcb, dze, gq,
                       M, Gr, c.codebook_size, q.d, q.dt, q.Nz, Tq, scale, T.freeze);
    {
        const dim3 cg(Gr * ((c.codebook_size + CBG_CODES - 1) / CBG_CODES));
        float* dcbw = G(h, q.cbraw);
        if (q.dt <= 8) hipLaunchKernelGGL(codebook_grad_kernel<8>, cg, dim3(256), 0, st, codes, bstride, gq, dcbw, M, Gr, c.codebook_size, q.dt, q.Nz, Tq);
        else if (q.dt <= 16) hipLaunchKernelGGL(codebook_grad_kernel<16>, cg, dim3(256), 0, st, codes, bstride, gq, dcbw, M, Gr, c.codebook_size, q.dt, q.Nz, Tq);
        else if (q.dt <= 32) hipLaunchKernelGGL(codebook_grad_kernel<32>, cg, dim3(256), 0, st, codes, bstride, gq, dcbw, M, Gr, c.codebook_size, q.dt, q.Nz, Tq);
        else if (q.dt <= 64) hipLaunchKernelGGL(codebook_grad_kernel<64>, cg, dim3(256), 0, st, codes, bstride, gq, dcbw, M, Gr, c.codebook_size, q.dt, q.Nz, Tq);
        else ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "codebook gradient: code dimension above 64");
    }
    // down-projection: ze = residual_frames . Wd^T   (Wd packed [Nz][Kq])
    ResidualGatherA rfr{Q.enc, Q.dec, q.Hq, W, q.Cp, Tq, c.overlap, M, FastDiv(Tq), FastDiv(q.Cp), FastDiv(q.Hq)};
    if ((rc = dw_launch(h, PlainA{dze, q.Nz, M}, rfr, M, q.Nz, q.Kq, G(h, q.wd), nullptr, part, st))) return rc;
    gemm_rows(dze, q.Nz, M, q.wdT, q.Kq, q.Nz, EpiPvqGrad{d_enc, ddec, q.Hq, W, q.Cp, Tq, c.overlap, FastDiv(Tq), FastDiv(q.Cp), FastDiv(q.Hq)}, st);
    return launch_ok("quant_bwd");
}

void add_inplace(float* dst, const float* src, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks_for((long long)(n / 4))), dim3(256), 0, st, dst, src, (long long)(n / 4));
}

}  // namespace

namespace {
int train_backward_impl(escx_handle_s* h, const float* d_wave, const float* d_recon_feat, const float* d_cm, const float* d_cb, float* grad_flat, hipStream_t st) {
    int rc = 0;
    TrainTape& T = *tape_of(h);
    if (!T.valid) ESCX_FAIL(ESCX_ERR_STATE, "escx_train_backward without a preceding escx_train_forward");
    T.valid = false;                                             // the tape is consumed (scratch overwrites nothing of it, but one backward per forward)
    const escx_config& c = h->cfg;
    const Shapes& s = T.shp;
    Arena& tp = h->tape;
    tp.used = T.fwd_mark;
    const int n = h->n, B = T.B;
    const int T2 = c.patch_t * s.W, F2 = c.patch_f * s.H0, out_len = c.hop_length * (T2 - 1);
    ESCX_HIP(hipMemsetAsync(h->garena, 0, h->wts.cap, st));
    ESCX_HIP(hipMemsetAsync(grad_flat, 0, h->flat_total * sizeof(float), st));
    Scratch top(&tp);
    // gradients of the encoder maps (two consumers each: the next encoder layer and a quantiser)
    std::vector<float*> d_enc(n);
    for (int i = 0; i < n; ++i) {
        const size_t nfl = (size_t)B * s.encH[i] * s.W * rup(c.h_dims[i], 16);
        d_enc[i] = top.take(nfl);
        if (!d_enc[i]) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        ESCX_HIP(hipMemsetAsync(d_enc[i], 0, nfl * sizeof(float), st));
    }
    // ---- inverse STFT + de-embedding ----
    const size_t tok0 = (size_t)B * s.H0 * s.W * h->C0p;
    float* gtok = top.take(tok0);                                // gradient of decoder.post_nn's output
    if (!gtok) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
    {
        Scratch sc(&tp);
        const size_t nspec = (size_t)B * T2 * c.in_dim * h->Fp;
        float* drspec = sc.take(nspec);
        float* dframes = sc.take((size_t)B * T2 * h->winP);
        float* part = sc.take(DW_PART_FLOATS);
        if (!part) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        if (d_recon_feat) pad_rows(d_recon_feat, drspec, (long long)B * T2 * c.in_dim, h->F, h->Fp, st);
        else ESCX_HIP(hipMemsetAsync(drspec, 0, nspec * sizeof(float), st));
        if (d_wave) {
            hipLaunchKernelGGL(istft_ola_bwd_kernel, dim3(blocks_for((long long)B * T2 * h->winP)), dim3(256), 0, st, d_wave, h->win2, dframes, B, T2,
                               h->winP, c.win_length, c.hop_length, h->left, h->n_fft / 2, out_len);
            PROF("B.istft", 2.0 * B * T2 * c.win_length * 2 * h->F, 0,
                 gemm_rows(dframes, h->winP, B * T2, h->idft_wT, c.in_dim * h->Fp, h->winP, EpiAccum{drspec, c.in_dim * h->Fp}, st));
        }
        // conv5x5 + pixel shuffle (in: tokens [B][H0][W][C0p], out: fine map), through the rank-18 structure of its output gradient
        // (train_kernels.h: deembed_p_kernel): dW and dX contract over 128 columns of P instead of the 288 channels of dY1
        const int Mt = B * s.H0 * s.W, Q = h->Q, K1 = 25 * h->C0p;
        if (Q * DEP_J > DEP_LD || c.in_dim * 9 > DEP_J) ESCX_FAIL(ESCX_ERR_UNSUPPORTED, "patch size / in_dim outside the de-embedding backward kernels");
        // Distinct-value columns (train_kernels.h, deembed_d_kernel): 48 instead of 128.  Needs the forward without the fine map (X from R) and the
        // 40 values to fit; ESCX_TRAIN_DEEMBED_SLOTS=0 keeps the 128-column P (A/B baseline).
        static const bool want_slots = [] { const char* e = getenv("ESCX_TRAIN_DEEMBED_SLOTS"); return !(e && e[0] == '0'); }();
        const int pf = c.patch_f, pt = c.patch_t;
        const bool slots = want_slots && T.composed && c.in_dim * (pf + 2) * (pt + 2) <= DEP_LD2;
        const int LD = slots ? DEP_LD2 : DEP_LD;
        float* P = sc.take((size_t)Mt * LD);
        float* R = sc.take((size_t)LD * K1 + LD);
        float* weff = sc.take((size_t)h->C0p * 25 * LD);
        if (!weff) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        if (slots) hipLaunchKernelGGL(deembed_d_kernel, dim3(blocks_for((long long)Mt * LD)), dim3(256), 0, st, drspec, P, B, s.H0, s.W, pf, pt, c.in_dim, h->Fp);
        else hipLaunchKernelGGL(deembed_p_kernel, dim3(blocks_for((long long)Mt * LD)), dim3(256), 0, st, drspec, P, B, s.H0, s.W, pf, pt, c.in_dim, h->Fp);
        ConvA ctok{T.post, s.H0, s.W, h->C0p, 5, 5, Mt};
        float* Rb = R + (size_t)LD * K1;
        PROF("B.dw_conv5", 2.0 * Mt * 25 * h->C0 * (slots ? c.in_dim * (pf + 2) * (pt + 2) : Q * c.in_dim * 9), 0,
             rc = slots ? (dw_launch_wide<3, 4, 1, 4>(h, PlainA{P, LD, Mt}, ctok, Mt, LD, K1, R, Rb, part, st))          // 48 x 256 workgroup tiles
                        : (dw_launch_wide<4, 4>(h, PlainA{P, LD, Mt}, ctok, Mt, LD, K1, R, Rb, part, st)));
        if (rc) return rc;
        hipLaunchKernelGGL(deembed_fold_dw_kernel, dim3(blocks_for((long long)Q * h->C0p * K1)), dim3(256), 0, st, R, h->dc2_w, G(h, h->dc1_w), Q, h->C0, h->C0p, K1, c.in_dim, pf, pt, slots);
        hipLaunchKernelGGL(deembed_fold_dw_kernel, dim3(blocks_for((long long)Q * h->C0p)), dim3(256), 0, st, Rb, h->dc2_w, G(h, h->dc1_b), Q, h->C0, h->C0p, 1, c.in_dim, pf, pt, slots);
        if (slots) hipLaunchKernelGGL(deembed_weff_slots_kernel, dim3(blocks_for((long long)h->C0p * 25 * LD)), dim3(256), 0, st, h->dc1_w, h->dc2_w, weff, h->C0, h->C0p, pf, pt, c.in_dim);
        else hipLaunchKernelGGL(deembed_weff_kernel, dim3(blocks_for((long long)h->C0p * 25 * LD)), dim3(256), 0, st, h->dc1_w, h->dc2_w, weff, Q, h->C0, h->C0p, c.in_dim);
        // conv3x3 weight gradient with the same P: only the Q diagonal blocks of P^T . Y1 (Y1 = the saved fine map viewed per coarse pixel) are
        // needed - one 20 x C0p contraction per sub-pixel q instead of the full 128 x Q*C0p product; db2 = the centre-tap column sums of P
        float* X = sc.take((size_t)Q * DEP_J * h->C0p);
        if (!X) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        if (T.composed) {               // no saved fine map: X from R, the conv5x5 weights and the column sums of P (train_kernels.h)
            ESCX_HIP(hipMemsetAsync(X, 0, (size_t)Q * DEP_J * h->C0p * sizeof(float), st));
            const int nj = c.in_dim * 9;
            PROF("B.dw_conv3", 2.0 * Q * nj * h->C0 * K1, 0,
                 hipLaunchKernelGGL(deembed_x_from_r_kernel, dim3(blocks_for((long long)Q * nj * h->C0p * 64)), dim3(256), 0, st, R, Rb, h->dc1_w, h->dc1_b, X,
                                    Q, h->C0, h->C0p, K1, nj, pf, pt, slots));
        } else {
        ShuffleA ysh{T.deemb, s.H0, s.W, h->C0p, c.patch_f, c.patch_t, Mt, FastDiv(s.H0 * s.W), FastDiv(s.W), FastDiv(h->C0p)};
        PROF("B.dw_conv3", 2.0 * Mt * Q * 9 * h->C0 * c.in_dim, 0, {
             for (int q = 0; q < Q && !rc; ++q)
                 rc = dw_launch(h, ColOffset<PlainA>{PlainA{P, LD, Mt}, q * DEP_J}, ColOffset<ShuffleA>{ysh, q * h->C0p}, Mt, DEP_J, h->C0p,
                                X + (size_t)q * DEP_J * h->C0p, nullptr, part, st); });
        if (rc) return rc;
        }
        hipLaunchKernelGGL(deembed_fold_dw2_kernel, dim3(blocks_for((long long)c.in_dim * 9 * h->C0p + c.in_dim)), dim3(256), 0, st, X, Rb, G(h, h->dc2_w),
                           G(h, h->dc2_b), Q, h->C0, h->C0p, c.in_dim, pf, pt, slots);
        ConvA cp{P, s.H0, s.W, LD, 5, 5, Mt};
        PROF("B.dx_conv5", 2.0 * Mt * 25 * h->C0 * (slots ? c.in_dim * (pf + 2) * (pt + 2) : Q * c.in_dim * 9), 0,
             gemm_any(cp, weff, Mt, h->C0p, 25 * LD, EpiStore{gtok, h->C0p, nullptr}, st, pick_bk(LD)));
    }
    // ---- decoder (post_nn, blocks n-2 .. 0) interleaved with the quantisers ----
    float* gcur = nullptr;
    Scratch dec_sc(&tp);
    size_t gdec_floats = tok0;                // running gradient of the decoder map entering the next (earlier) stage
    for (int li = n; li < 2 * n; ++li) gdec_floats = std::max(gdec_floats, (size_t)B * layer_H(h, s, li) * s.W * h->layers[li].Cp);
    float* gdec = dec_sc.take(gdec_floats);
    if (!gdec) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
    {
        Scratch sc(&tp);
        if ((rc = layer_bwd(h, h->layers[2 * n - 1], T.layers[2 * n - 1], gtok, B, s.W, grad_flat, sc, &gcur, st))) return rc;
        const Layer& L = h->layers[2 * n - 1];
        ESCX_HIP(hipMemcpyAsync(gdec, gcur, (size_t)B * T.layers[2 * n - 1].H * s.W * L.Cp * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int i = n - 2; i >= 0; --i) {
        const Layer& L = h->layers[n + i];
        const LayerTape& LT = T.layers[n + i];
        const size_t nin = (size_t)B * LT.H * s.W * L.Cp;
        {
            Scratch sc(&tp);
            if ((rc = layer_bwd(h, L, LT, gdec, B, s.W, grad_flat, sc, &gcur, st))) return rc;       // gcur = gradient of the refined map (block input)
            ESCX_HIP(hipMemcpyAsync(gdec, gcur, nin * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        // the refined map = dec + residual_q: the +dec path keeps gdec as it is; the quantiser adds to d_enc and subtracts its residual gradient from gdec
        if (T.q[i + 1].transmit) {           // (every read of the map gradient inside quant_bwd precedes the epilogue that updates it: one buffer serves as both)
            Scratch sc(&tp);
            if ((rc = quant_bwd(h, T, i + 1, gdec, d_enc[n - 1 - i], gdec, d_cm, d_cb, sc, st))) return rc;
        }
    }
    {   // stream 0: the refined map is residual_q alone (dec = 0.0)
        Scratch sc(&tp);
        if ((rc = quant_bwd(h, T, 0, gdec, d_enc[n - 1], nullptr, d_cm, d_cb, sc, st))) return rc;
    }
    // ---- encoder ----
    for (int i = n - 2; i >= 0; --i) {
        Scratch sc(&tp);
        const Layer& L = h->layers[1 + i];
        if ((rc = layer_bwd(h, L, T.layers[1 + i], d_enc[i + 1], B, s.W, grad_flat, sc, &gcur, st))) return rc;
        add_inplace(d_enc[i], gcur, (size_t)B * s.encH[i] * s.W * L.Cp, st);
    }
    {
        Scratch sc(&tp);
        if ((rc = layer_bwd(h, h->layers[0], T.layers[0], d_enc[0], B, s.W, grad_flat, sc, &gcur, st))) return rc;
        // patch embedding: LN + strided conv as a GEMM over gathered patches (scale.py:42-50); the input is data: no dX
        float* dpre = sc.take(tok0);
        float* lnpart = sc.take(LN_PART_FLOATS);
        float* part = sc.take(DW_PART_FLOATS);
        if (!part) ESCX_FAIL(ESCX_ERR_STATE, "training tape too small (backward scratch)");
        const int Mt = B * s.H0 * s.W;
        ln_bwd(0, T.pe_pre, gcur, h->pe_g, nullptr, nullptr, dpre, G(h, h->pe_g), G(h, h->pe_beta), s.H0 * s.W, s.H0 * s.W, s.H0 * s.W, Mt, h->C0, h->C0p, lnpart, st);
        PatchA pa{T.spec, s.T, c.in_dim * h->Fp, h->Fp, s.H0, s.W, c.patch_f, c.patch_t, c.in_dim * c.patch_f * c.patch_t, Mt};
        if ((rc = dw_launch(h, PlainA{dpre, h->C0p, Mt}, pa, Mt, h->C0p, h->Kpe, G(h, h->pe_w), G(h, h->pe_b), part, st))) return rc;
    }
    // ---- packed gradients -> flat reference layout ----
    if (!h->grad_seg && !h->grad_regions.empty()) {              // one table for all primary layouts (344 launches per step before)
        std::vector<long long> seg;
        long long tot = 0;
        for (auto& r : h->grad_regions) { seg.push_back(tot); seg.push_back((long long)r.first); tot += (long long)r.second; }
        ESCX_HIP(hipMalloc((void**)&h->grad_seg, seg.size() * sizeof(long long)));
        ESCX_HIP(hipMemcpy(h->grad_seg, seg.data(), seg.size() * sizeof(long long), hipMemcpyHostToDevice));
        h->grad_nseg = (int)h->grad_regions.size(); h->grad_seg_total = tot;
    }
    if (h->grad_seg)
        hipLaunchKernelGGL(scatter_grads_kernel, dim3(blocks_for(h->grad_seg_total)), dim3(256), 0, st, h->garena, h->gmap, grad_flat, h->grad_seg, h->grad_nseg,
                           h->grad_seg_total);
    return launch_ok("train_backward");
}
}  // namespace

extern "C" int escx_train_backward(escx_handle h, const float* d_wave, const float* d_recon_feat, const float* d_cm, const float* d_cb,
                                   float* grad_flat, void* stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!grad_flat) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null gradient buffer");
    TrainRoot& R = *root_of(h);
    if (!R.valid) ESCX_FAIL(ESCX_ERR_STATE, "escx_train_backward without a preceding escx_train_forward");
    R.valid = false;
    hipStream_t st = (hipStream_t)stream;
    if (R.parts == 1) { R.cur = &R.single; return train_backward_impl(h, d_wave, d_recon_feat, d_cm, d_cb, grad_flat, st); }
    const escx_config& c = h->cfg;
    const Shapes& s1 = R.part[0].shp;
    const size_t per_wave = (size_t)c.hop_length * (c.patch_t * s1.W - 1), per_recon = (size_t)c.patch_t * s1.W * c.in_dim * h->F;
    ESCX_HIP(hipEventRecord(R.ev_fork, st));
    for (int p = 1; p < R.parts; ++p) {
        if (!R.garena[p]) ESCX_HIP(hipMalloc((void**)&R.garena[p], h->wts.cap));
        if (!R.gflat[p]) ESCX_HIP(hipMalloc((void**)&R.gflat[p], h->flat_total * sizeof(float)));
        ESCX_HIP(hipStreamWaitEvent(R.aux[p], R.ev_fork, 0));
    }
    for (int p = 0; p < R.parts; ++p) {
        const int b0 = R.first[p];
        std::swap(h->tape, R.arena[p]);
        if (p) std::swap(h->garena, R.garena[p]);
        R.cur = &R.part[p];
        rc = train_backward_impl(h, d_wave ? d_wave + b0 * per_wave : nullptr, d_recon_feat ? d_recon_feat + b0 * per_recon : nullptr, d_cm ? d_cm + b0 : nullptr,
                                 d_cb ? d_cb + b0 : nullptr, p ? R.gflat[p] : grad_flat, p ? R.aux[p] : st);
        std::swap(h->tape, R.arena[p]);
        if (p) std::swap(h->garena, R.garena[p]);
        R.cur = &R.single;
        if (rc) break;                  // the aux streams are joined below on the error path too
    }
    // d loss / d parameter = ((part 0 + part 1) + part 2) + ... (fixed order)
    for (int p = 1; p < R.parts; ++p) {
        ESCX_HIP(hipEventRecord(R.ev_join[p], R.aux[p]));
        ESCX_HIP(hipStreamWaitEvent(st, R.ev_join[p], 0));
        if (rc) continue;
        hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks_for((long long)h->flat_total / 4 + 1)), dim3(256), 0, st, grad_flat, R.gflat[p], (long long)(h->flat_total / 4));
        if (h->flat_total % 4) hipLaunchKernelGGL(add_tail_kernel, dim3(1), dim3(4), 0, st, grad_flat, R.gflat[p], (long long)(h->flat_total / 4 * 4), (long long)h->flat_total);
    }
    if (rc) return rc;
    return launch_ok("train_backward");
}

// ---------------------------------------------------------------------------------------------------------
// losses (generator_loss.py) with their gradients
// ---------------------------------------------------------------------------------------------------------
extern "C" int escx_stft_loss(const float* raw_feat, const float* recon_feat, int B, int64_t per_clip, float* loss, float* d_recon, void* stream) {
    if (!raw_feat || !recon_feat || !loss || B < 1 || per_clip < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long long per = (long long)per_clip;
    const int bpc = (int)std::min<long long>(64, (per + 255) / 256);
    float* part = stream_scratch(st, 0, (size_t)B * bpc);
    if (!part) ESCX_FAIL(ESCX_ERR_HIP, "scratch allocation failed");
    hipLaunchKernelGGL(stft_loss_kernel, dim3(bpc, B), dim3(256), 0, st, raw_feat, recon_feat, part, d_recon, per, bpc, 1.0f / (float)per);
    hipLaunchKernelGGL(row_sum_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, part, bpc, loss, B, 0, 1.0f);
    return launch_ok("stft_loss");
}

namespace {
const int MEL_WINDOWS[7] = {32, 64, 128, 256, 512, 1024, 2048};
const int MEL_BINS[7] = {5, 10, 20, 40, 80, 160, 320};
struct MelScale { int w, hop, F, Fq, n_mels, Mp; float *D, *DT, *fb, *fbT; };
struct MelState { MelScale sc[7]; float* base = nullptr; int sr = 0; };
MelState* g_mel[64] = {nullptr};            // per device: DFT / mel filterbank matrices of the 7 resolutions (constants, built once)

int build_mel(int dev, int sr, MelState** out) {
    if (dev < 0 || dev >= 64) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "device index out of range");
    if (g_mel[dev] && g_mel[dev]->sr == sr) { *out = g_mel[dev]; return 0; }
    if (g_mel[dev]) { (void)hipDeviceSynchronize(); (void)hipFree(g_mel[dev]->base); delete g_mel[dev]; g_mel[dev] = nullptr; }
    MelState* ms = new MelState();
    std::vector<float> host;
    std::vector<size_t> offs;
    auto alloc = [&](size_t n) { size_t o = (host.size() + 63) / 64 * 64; host.resize(o + n, 0.f); return o; };
    size_t oD[7], oDT[7], ofb[7], ofbT[7];
    for (int i = 0; i < 7; ++i) {
        MelScale& m = ms->sc[i];
        m.w = MEL_WINDOWS[i]; m.hop = m.w / 4; m.F = m.w / 2 + 1; m.Fq = rup(m.F, 16); m.n_mels = MEL_BINS[i]; m.Mp = rup(m.n_mels, 16);
        oD[i] = alloc((size_t)2 * m.Fq * m.w); oDT[i] = alloc((size_t)m.w * 2 * m.Fq); ofb[i] = alloc((size_t)m.Mp * m.Fq); ofbT[i] = alloc((size_t)m.Fq * m.Mp);
        for (int f = 0; f < m.F; ++f) for (int k = 0; k < m.w; ++k) {
            const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * k / m.w);           // hann, periodic (torch.hann_window default)
            const double ang = 2.0 * M_PI * (double)((long long)f * k % m.w) / m.w;
            const float re = (float)(win * std::cos(ang)), im = (float)(-win * std::sin(ang));
            host[oD[i] + (size_t)f * m.w + k] = re; host[oD[i] + (size_t)(m.Fq + f) * m.w + k] = im;
            host[oDT[i] + (size_t)k * 2 * m.Fq + f] = re; host[oDT[i] + (size_t)k * 2 * m.Fq + m.Fq + f] = im;
        }
        // torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale="htk")
        auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
        auto mel2hz = [](double mm) { return 700.0 * (std::pow(10.0, mm / 2595.0) - 1.0); };
        const double mlo = hz2mel(0.0), mhi = hz2mel(sr / 2.0);
        std::vector<double> fp(m.n_mels + 2);
        for (int j = 0; j < m.n_mels + 2; ++j) fp[j] = mel2hz(mlo + (mhi - mlo) * j / (m.n_mels + 1));
        for (int f = 0; f < m.F; ++f) {
            const double fr = (double)(sr / 2) * f / (m.F - 1);
            for (int j = 0; j < m.n_mels; ++j) {
                const double down = (fr - fp[j]) / (fp[j + 1] - fp[j]), up = (fp[j + 2] - fr) / (fp[j + 2] - fp[j + 1]);
                const float v = (float)std::max(0.0, std::min(down, up));
                host[ofb[i] + (size_t)j * m.Fq + f] = v; host[ofbT[i] + (size_t)f * m.Mp + j] = v;
            }
        }
    }
    ESCX_HIP(hipMalloc((void**)&ms->base, host.size() * sizeof(float)));
    ESCX_HIP(hipMemcpy(ms->base, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    for (int i = 0; i < 7; ++i) { ms->sc[i].D = ms->base + oD[i]; ms->sc[i].DT = ms->base + oDT[i]; ms->sc[i].fb = ms->base + ofb[i]; ms->sc[i].fbT = ms->base + ofbT[i]; }
    ms->sr = sr;
    g_mel[dev] = ms;
    *out = ms;
    return 0;
}
}  // namespace

// MelSpectrogramLoss (generator_loss.py:37-74): 7 resolutions, L1 on mel magnitudes + L1 on log10(mel^2); d_recon (B, L) optional.
// STFTs are windowed-DFT GEMMs (FrameA loader with reflect padding), the mel projection another GEMM; nothing leaves the device.
extern "C" int escx_mel_loss(const float* raw_wave, const float* recon_wave, int B, int L, int sample_rate, float* loss, float* d_recon, void* stream) {
    if (!raw_wave || !recon_wave || !loss || B < 1 || sample_rate < 2) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    if (L <= 1024) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_samples=%d too short for the 2048-sample mel window (reflect padding)", L);
    int dev = 0;
    ESCX_HIP(hipGetDevice(&dev));
    MelState* msp = nullptr;
    int rc = build_mel(dev, sample_rate, &msp);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const MelState& ms = *msp;
    // scratch: sized for the worst scale
    size_t spec_max = 0, mag_max = 0, mel_max = 0, fr_max = 0;
    for (int i = 0; i < 7; ++i) {
        const MelScale& m = ms.sc[i];
        const size_t rows = (size_t)B * (1 + L / m.hop);
        spec_max = std::max(spec_max, rows * 2 * m.Fq); mag_max = std::max(mag_max, rows * m.Fq); mel_max = std::max(mel_max, rows * m.Mp);
        fr_max = std::max(fr_max, rows * m.w);
    }
    const int bpc = 64;
    const size_t total = 2 * spec_max + 3 * mag_max + 3 * mel_max + fr_max + (size_t)B * bpc + 1024;
    float* buf = stream_scratch(st, 1, total);
    if (!buf) ESCX_FAIL(ESCX_ERR_HIP, "scratch allocation of %zu floats failed", total);
    float* spx = buf; float* spy = spx + spec_max; float* mgx = spy + spec_max; float* mgy = mgx + mag_max; float* dmg = mgy + mag_max;
    float* mlx = dmg + mag_max; float* mly = mlx + mel_max; float* gml = mly + mel_max; float* dfr = gml + mel_max; float* part = dfr + fr_max;
    for (int i = 0; i < 7; ++i) {
        const MelScale& m = ms.sc[i];
        const int Tm = 1 + L / m.hop; const long long rows = (long long)B * Tm;
        for (int side = 0; side < 2; ++side) {
            const float* wv = side ? recon_wave : raw_wave;
            float* sp = side ? spy : spx; float* mg = side ? mgy : mgx; float* ml = side ? mly : mlx;
            gemm_frames(wv, B, L, Tm, m.hop, -m.w / 2, m.D, 2 * m.Fq, m.w, sp, st);
            hipLaunchKernelGGL(complex_mag_kernel, dim3(blocks_for(rows * m.Fq)), dim3(256), 0, st, sp, mg, rows, m.F, m.Fq, m.Fq);
            gemm_rows(mg, m.Fq, (int)rows, m.fb, m.Mp, m.Fq, EpiStore{ml, m.Mp, nullptr}, st);
        }
        hipLaunchKernelGGL(mel_l1_kernel, dim3(bpc, B), dim3(256), 0, st, mlx, mly, part, d_recon ? gml : nullptr, Tm, m.n_mels, m.Mp, bpc,
                           1.0f / ((float)m.n_mels * Tm), 1e-5f);
        hipLaunchKernelGGL(row_sum_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, part, bpc, loss, B, i > 0, 1.0f);
        if (d_recon) {
            gemm_rows(gml, m.Mp, (int)rows, m.fbT, m.Fq, m.Mp, EpiStore{dmg, m.Fq, nullptr}, st);
            hipLaunchKernelGGL(complex_mag_bwd_kernel, dim3(blocks_for(rows * m.Fq)), dim3(256), 0, st, spy, dmg, spx, rows, m.F, m.Fq, m.Fq);     // spx is free: reuse as d spec
            gemm_rows(spx, 2 * m.Fq, (int)rows, m.DT, m.w, 2 * m.Fq, EpiStore{dfr, m.w, nullptr}, st);
            hipLaunchKernelGGL(frames_bwd_kernel, dim3(blocks_for((long long)B * L)), dim3(256), 0, st, dfr, d_recon, B, L, Tm, m.hop, m.w, m.w, i > 0);
        }
    }
    return launch_ok("mel_loss");
}

extern "C" int escx_scale_rows(const float* x, const float* g, float* out, int rows, int64_t per_row, void* stream) {
    if (!x || !g || !out || rows < 0 || per_row < 0) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    const long long n = (long long)rows * per_row;
    if (n) hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, g, out, (long long)per_row, n);
    return launch_ok("scale_rows");
}

// ---------------------------------------------------------------------------------------------------------
// optimiser on flat buffers (trainer_no_adv.py:116-117: clip_grad_norm_(0.5) + AdamW step)
// ---------------------------------------------------------------------------------------------------------
// norm_out_dev[0] = global L2 norm, norm_out_dev[1] = clip coefficient (torch.nn.utils.clip_grad_norm_ semantics); needs 2 + 1024 floats of scratch after it
extern "C" int escx_grad_norm_clip(const float* grad_flat, int64_t n, float max_norm, float* norm_out_dev, void* stream) {
    if (!grad_flat || !norm_out_dev || n < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)std::min<long long>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, grad_flat, (long long)n, norm_out_dev + 2);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, st, norm_out_dev + 2, blocks, max_norm, norm_out_dev);
    return launch_ok("grad_norm_clip");
}
// torch.optim.AdamW step t (1-based) on flat parameter / gradient / moment buffers; clip_dev (optional) = output of escx_grad_norm_clip
extern "C" int escx_adamw_step(float* param_flat, const float* grad_flat, float* exp_avg, float* exp_avg_sq, int64_t n, int step, double lr, double beta1,
                               double beta2, double eps, double weight_decay, const float* clip_dev, void* stream) {
    if (!param_flat || !grad_flat || !exp_avg || !exp_avg_sq || n < 1 || step < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "bad argument");
    // torch.optim.AdamW evaluates the bias corrections, lr / bc1, sqrt(bc2), 1 - beta and 1 - lr * wd in double (python floats) and hands its
    // kernels fp32 scalars: the same here (fp32 1 - 0.999f alone is off by 1.3e-5 relative)
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for((long long)n)), dim3(256), 0, (hipStream_t)stream, param_flat, grad_flat, exp_avg, exp_avg_sq, (long long)n,
                       clip_dev, (float)(1.0 - lr * weight_decay), (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(lr / bc1), (float)std::sqrt(bc2));
    return launch_ok("adamw_step");
}
