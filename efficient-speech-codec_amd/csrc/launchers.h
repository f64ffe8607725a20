// Host-callable launchers (one per kernel group); implemented in gemm_swin.hip / gemm_misc.hip / kernels_misc.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "tune_env.h"

namespace escx {

// ---- Swin block / scale-change linears (gemm_swin.hip) ----
void gemm_qkv(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* bias,
              int nq, float scale, hipStream_t s);
void gemm_proj_scatter(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* shortcut,
                       const float* bias, const int* map, int slots, int tokens, hipStream_t s);
void gemm_gelu(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* bias, hipStream_t s);
void gemm_residual(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* bias,
                   const float* res, hipStream_t s);
void gemm_store(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, int ldo, const float* bias, hipStream_t s);
void gemm_split(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, int H, int Wd, int C2p, hipStream_t s);

// widest padded row (CP * TM) for which the fused MLP is compiled for 4 waves per SIMD (128 VGPRs); 3 waves up to 192
#ifndef ESCX_MLP_OCC4
#define ESCX_MLP_OCC4 96
#endif
// windows per wave of the fused attention: two up to this padded width (they share every weight fragment), one above
#ifndef ESCX_ATTN_TMW2_MAX
#define ESCX_ATTN_TMW2_MAX 48
#endif
constexpr int attn_windows_per_wave(int Cp) { return Cp <= ESCX_ATTN_TMW2_MAX ? 2 : 1; }

// ---- fused register-resident Swin kernels (fused_swin.hip); return -1 when the width is not instantiated ----
constexpr int ESCX_COMB_UNSUPPORTED = -3;     // "this kernel has no instantiation for the requested fused form": nothing was launched
// PatchSplit folded into the epilogue of a layer's LAST fused MLP (fused_mlp.h, SPLIT instantiations): LayerNorm(C) + Linear(C -> 2 C') + two-row scatter of
// x + mlp(x) from registers; x itself is not written.  wf: Layer::sub_wf (fragment order), NT = 2 * C'p / 16 output tiles (even), out = the (2H, W, C'p) map.
struct MlpSplit { const float* wf; const float* gamma; const float* beta; float* out; int NT, H, W, C2p; };
int mlp_fused(float* x, int M, int C, int Cp, const float* gamma, const float* beta, const float* w1f, const float* b1,
              const float* w2f, const float* b2, const float* wcf, int hiddenP, int variant, int* hs_io, float* partial, hipStream_t s,
              float* out = nullptr,       // out != nullptr: x untouched, x + mlp(x) goes to out (not with a hidden split)
              int* tickets = nullptr, int n_tickets = 0, bool* combined = nullptr,     // hidden split: arrival counters (zeroed) -> *combined = the launch did the combine itself
              const MlpSplit* split = nullptr);   // ESCX_COMB_UNSUPPORTED (nothing launched) when this width / variant has no SPLIT instantiation
unsigned long long* debug_trace_buffer();      // device buffer set by escx_debug_mlp_trace (tuning builds: in-kernel phase stamps), or nullptr
// Fused MLP on the bf16 matrix cores with fp32 operands split into three bf16 terms (fused_mlp_x3.h).  mlp_x3_pack builds the split weight image from the row-major
// fp32 weights (w1 [hiddenP][Cp], w2 [Cp][hiddenP]); mlp_x3_bytes = its size.  -1: width not instantiated.
// nt: terms per operand - 3 = bf16 (exact split, six cross products), 2 = fp16 with per-matrix power-of-two scales (three cross products); image and kernel must agree
size_t mlp_x3_bytes(int Cp, int hiddenP, int nt = 3);
int mlp_x3_pack(const float* w1, const float* w2, void* image, int Cp, int hiddenP, hipStream_t s, int nt = 3, const float* gamma = nullptr, const float* beta = nullptr,
                const float* b1 = nullptr, int C = 0);      // nt = 2 needs ln2's gamma / beta, b1 and the unpadded width (range rule of split_terms.h)
struct MlpSplit;
int mlp_x3(float* x, int M, int C, int Cp, const float* gamma, const float* beta, const float* b1, const float* b2, const void* image, int hiddenP, int nw, int* hs_io, float* partial,
           hipStream_t s, const MlpSplit* split = nullptr, int nt = 3, float* out = nullptr);      // split: PatchSplit in the epilogue, split->wf = image of mlp_x3_split_pack
size_t mlp_x3_split_bytes(int Cp, int Np);
int mlp_x3_split_pack(const float* wf, void* image, int Cp, int Np, hipStream_t s);
void rows_combine(float* dst, const float* src, const float* partial, const float* bias, long long M, int Cp, int n, hipStream_t s);

// LN + linear for PatchMerge (segs = 2, map gives the two source rows) / PatchSplit (segs = 1, split = 1: pixel-shuffled store)
// Pending combine of a hidden-split MLP (fused_mlp.h): the consumer forms x + (((P0 + P1) + ...) + bias) while it loads its rows.  Kernels without a
// combine-on-load instantiation return ESCX_COMB_UNSUPPORTED without launching: the caller then runs rows_combine and calls again without it.
struct CombineOnLoad { const float* partial; const float* bias; long long stride; int n; };
int rowgemm_fused(int segs, const float* x, float* out, const float* gamma, const float* beta, const float* wf, const int* map, int M,
                  int rows_per_clip, int src_rows_per_clip, int C, int Cp, int Np, int split, int H, int W, int C2p, hipStream_t s,
                  const CombineOnLoad* comb = nullptr,
                  const void* x3_wf = nullptr, int x3_nt = 3);      // split weight stream (rowgemm_x3_pack, x3_nt terms: 3 = bf16, 2 = fp16 + scales), or nullptr = fp32 MFMA
size_t rowgemm_x3_bytes(int KP, int Np);
int rowgemm_x3_pack(const float* wf, void* image, int KP, int Np, hipStream_t s, int nt = 3, const float* gamma = nullptr, const float* beta = nullptr, int C = 0);   // gamma / beta over KP channels, C real ones: the LayerNorm whose output is split (nt = 2 range rule)
void loss_reduce(const float* terms, int n_slots, int G, int M, int Tq, float* out, hipStream_t s);     // per-clip commitment loss, fixed summation order
void mlp_set_trace(unsigned long long* p);
int test_fastdiv(int n, int d);       // host evaluation of gemm_engine.h FastDiv (gemm_misc.hip)
// halo-tiled composed de-embedding (fused_deembed.h); -1 when the width / output count is not instantiated
int deembed7_fused(const float* tok, int B, int H, int W, int Cp, const float* wfrag, const float* bias, float* out, int pf, int pt,
                   int in_dim, int Fp, hipStream_t s, const void* x2_image = nullptr);      // x2_image: two-term fp16 stream (deembed7_x2_pack), Cp = 48 only
size_t deembed7_x2_image_bytes(int Cp);
int deembed7_x2_pack(const float* wfrag, void* image, int Cp, hipStream_t s);
// mode: 0 one head (<=16 dims) per tile, 1 two heads (<=8 dims) per tile, 2 one head (<=32 dims) over two tiles
int attn_fused(const float* src, float* dst, int Cp, int C, int mode, int n_groups, const float* gamma, const float* beta,
               const float* wf, const float* bqkv, const float* bias_tab, const float* bproj, const int* map, int slots, int tokens,
               int n_windows, int nWh, int nWw, int shifted, float scale, int nw, int* gs_io, float* partial, int rows, hipStream_t s,
               const CombineOnLoad* comb = nullptr, const struct AttnTape* tape = nullptr,
               const void* x3_wf = nullptr, int x3_pairs = 0);     // split (3 x bf16) weight stream of this block (attn_x3_pack; pairs: its pair-order form), nullptr = fp32 MFMA
size_t attn_x3_bytes(int Cp, int mode, int n_groups);
int attn_x3_pack(const float* waf, void* image, int Cp, int mode, int n_groups, hipStream_t s, int pairs = 0, const float* gamma = nullptr, const float* beta = nullptr, int C = 0, const float* bqkv = nullptr);   // norm1's gamma / beta and the tile biases (pairs == 2: range rule; the output projection in two-term form too)
// training forward: the fused attention also writes what the backward reads (fused_attn.h, TAPE); returns ESCX_COMB_UNSUPPORTED when the width has
// no TAPE instantiation (the caller runs the unfused sequence)
struct AttnTape { float* xn; float* qkv; float* o; int ldq, ldo, hdp, nH; };

// ---- everything else that is a contraction (gemm_misc.hip) ----
void gemm_frames(const float* wave, int B, int L, int T, int hop, int off, const float* W, int Np, int Kp, float* out, hipStream_t s);
void gemm_patch(const float* spec, int B, int T, int in_dim, int Fp, int H, int Wd, int pf, int pt, const float* W, int Np, int Kp,
                float* out, const float* bias, hipStream_t s);
void gemm_conv_deembed1(const float* x, int B, int H, int Wd, int Cp, const float* W, int Np, float* out, const float* bias,
                        int pf, int pt, hipStream_t s);
void gemm_conv_spec(const float* x, int B, int T, int F, int Cp, const float* W, float* out, const float* bias, int Fp, int in_dim,
                    hipStream_t s);
// composed de-embedding (conv3x3 o pixel-shuffle o conv5x5 folded into one 7x7 convolution with in_dim*pf*pt outputs)
void gemm_deembed_composed(const float* x, int B, int H, int Wd, int Cp, const float* W, float* out, const float* bias, int pf, int pt,
                           int in_dim, int Fp, hipStream_t s);
int deembed_border(const float* x, const float* wv, const float* bv, float* out, int B, int H, int W, int C, int Cp, int pf, int pt,
                   int in_dim, int Fp, hipStream_t s);
void gemm_pvq_down(const float* enc, const float* dec, int B, int Hq, int Wd, int Cp, int ov, const float* W, int Np, int Kp,
                   float* zpart, int splits, hipStream_t s);
int pvq_down_splits(int M, int Kp, int Cp);
void gemm_pvq_up(const long long* codes, long long bstride, const float* cbraw, int G, int Ksz, int dt, int B, int Hq, int Wd, int Cp,
                 int ov, const float* W, int Np, int Kp, const float* dec, float* out, hipStream_t s);

// ---- non-GEMM kernels (kernels_misc.hip) ----
// mode 0: identity rows; 1: window gather (pad rows -> zeros after norm); 2: merge gather (2 segments)
void ln_rows(int mode, const float* src, float* dst, const float* gamma, const float* beta, const int* map, int rows_per_clip,
             int src_rows_per_clip, int total_rows, int C, int Cp, hipStream_t s);
int window_attention(const float* qkv, const float* bias, float* out, int total_windows, int nH, int hdp, int ldq, int ldo, int nWh,
                     int nWw, int shifted, hipStream_t s);
// any window size (kernels.h window_attention_any_kernel): the fallback for window_size != 4; bias [head][ws^2][ws^2], shift = 0 or ws / 2
int window_attention_any(const float* qkv, const float* bias, float* out, int total_windows, int ws, int nH, int hd, int hdp, int ldq, int ldo, int nWh, int nWw,
                         int shift, hipStream_t s);
// One product-VQ stream in ONE launch (fused_pvq.h): frame + residual + down-projection (split-K slices = waves, added in slice order in LDS) + normalise +
// codebook search + de-quantise + up-projection + un-frame + add.  out == nullptr: codes only; out may alias dec.  -1: geometry not covered.
// wdf: down-projection in fragment order; tab / gq: de-quantisation table and float4 -> group map (escx_internal.h Quant), tab == nullptr: up-projection on the MFMA
int pvq_fused(const float* enc, const float* dec, int B, int Hq, int Wd, int Cp, int ov, const float* wdf, int Np, int Kq, int splits, int bk,
              const float* cbn, const float* c2, const float* cbraw, int G, int Ksz, int d, int dt, const float* wup, const float* tab, const float* gq,
              float* out, long long* codes, long long bstride, float* loss, float loss_scale, int l2norm, hipStream_t s);
// out = dec + tab[(h, ov * code_g + o)][c] (g = group of element (o, h, c)): the de-quantise + up-projection + un-frame + add of one stream as a table-row add
void pvq_tab_add(const long long* codes, long long bstride, const float* tab, const float* gq, int G, int Ksz, int B, int Hq, int Wd, int Cp, int ov,
                 const float* dec, float* out, hipStream_t s);
// specialised PVQ framing + residual + down-projection (split-K partial sums, kernels.h); -1: geometry not covered, the caller falls back to gemm_pvq_down
int pvq_down(const float* enc, const float* dec, int B, int Hq, int Wd, int Cp, int ov, const float* W, int Np, int Kp, float* zpart, int splits,
             int bk, hipStream_t s);
int pvq_down_bk(int Cp);        // the K step the engine would use for this map width (fixes the slice boundaries)
// specialised PVQ de-quantise + up-project + un-frame + add (kernels.h); -1: geometry not covered, the caller falls back to gemm_pvq_up
int pvq_up(const long long* codes, long long bstride, const float* cbraw, int G, int Ksz, int dt, int B, int Hq, int Wd, int Cp, int ov,
           const float* W, int Np, int Kp, const float* dec, float* out, hipStream_t s);
int pvq_search(const float* zpart, int splits, int M, int ldz, const float* cbn, const float* c2, const float* cbraw, int G, int Ksz,
               int d, int dt, int Tq, long long* codes, long long bstride, float* loss, float loss_scale, int l2norm, hipStream_t s);
void istft_ola(const float* frames, const float* win2, float* wave, int B, int T, int ldf, int win, int hop, int left, int half,
               int out_len, hipStream_t s);
void pad_rows(const float* src, float* dst, long long rows, int C, int Cp, hipStream_t s);
void unpad_rows(const float* src, float* dst, long long rows, int C, int Cp, hipStream_t s);
void codes_pack10(const long long* in, unsigned char* out, long long n, hipStream_t s);
void codes_unpack10(const unsigned char* in, long long* out, long long n, hipStream_t s);
void test_math(const float* x, float* y, long long n, int which, hipStream_t s);
void test_copy_rows(const float* src, float* dst, long long rows, int Cp, hipStream_t s);
void codes_narrow(const long long* in, short* out, long long n, hipStream_t s);
void codes_widen(const short* in, long long* out, long long n, hipStream_t s);

}  // namespace escx
