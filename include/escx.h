/*
 * escx.h -- C ABI of the MI355X-native ESC encode/decode hot path (libescx.so, HIP / gfx950).
 *
 * The reference (yzGuu830/efficient-speech-codec) is pure Python; it has no FFI layer.  The boundary
 * this library replaces is the body of `esc.ESC.encode/.decode/.forward` (esc/models/codecs.py:30-94)
 * and the modules underneath.  Each entry point below names the reference function it stands for.
 * The reference-side binding a maintainer would add is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative escx_status on failure; the message for the
 *     calling thread is available from escx_last_error().  Nothing throws across the ABI.
 *   - `*_dev` pointers are device pointers owned by the caller (e.g. torch `tensor.data_ptr()`);
 *     `stream` is a hipStream_t passed as void* (0 = the null stream).  Calls are asynchronous with
 *     respect to the host unless stated otherwise.
 *   - parameters are uploaded once from HOST fp32 buffers under the reference's state_dict key names
 *      and packed into MFMA-friendly padded layouts owned by the handle.
 *   - the handle owns one workspace, sized by escx_reserve(); a handle serves one stream at a time.
 *   - stage-level entry points take and return tensors in the REFERENCE layouts (unpadded), so they can
 *     be compared one-to-one with the reference functions; the whole-path calls keep activations in the
 *     internal padded layouts between kernels.
 */
#ifndef ESCX_H
#define ESCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESCX_MAX_SCALES 8

typedef enum {
    ESCX_OK = 0,
    ESCX_ERR_INVALID_ARG = -1,      /* bad shape / null pointer / unknown key             */
    ESCX_ERR_UNSUPPORTED = -2,      /* configuration outside what the kernels implement   */
    ESCX_ERR_HIP = -3,              /* a HIP runtime call failed                          */
    ESCX_ERR_STATE = -4,            /* e.g. parameters not finalised                      */
    ESCX_ERR_ASSERT = -5            /* a reference assertion would have fired             */
} escx_status;

/* Mirrors the kwargs of esc.ESC.__init__ (esc/models/codecs.py:11-18). */
typedef struct {
    int32_t in_dim;                         /* 2                                           */
    int32_t in_freq;                        /* 192  -> n_fft = 2*(in_freq-1)               */
    int32_t n_scales;                       /* len(h_dims) = 6                             */
    int32_t h_dims[ESCX_MAX_SCALES];        /* 45,72,96,144,192,384                        */
    int32_t max_streams;                    /* 6                                           */
    int32_t win_length;                     /* int(win_len*sr*1e-3) = 320                  */
    int32_t hop_length;                     /* int(hop_len*sr*1e-3) = 80                   */
    int32_t patch_f, patch_t;               /* 3, 2                                        */
    int32_t swin_heads[ESCX_MAX_SCALES];    /* encoder order: 3,6,12,24,24                 */
    int32_t swin_depth;                     /* 2 (Base) / 4 (Large)                        */
    int32_t window_size;                    /* 2 .. 16; 4 = fused kernels, other sizes: unfused fallback, inference only */
    float   mlp_ratio;                      /* 4.0                                         */
    int32_t overlap;                        /* 2                                           */
    int32_t group_size;                     /* 3                                           */
    int32_t codebook_size;                  /* 1024                                        */
    int32_t codebook_dims[ESCX_MAX_SCALES]; /* per stream                                  */
    int32_t l2norm;                         /* 1                                           */
} escx_config;

typedef struct escx_handle_s* escx_handle;

/* ---- lifetime ------------------------------------------------------------------------------- */
const char* escx_last_error(void);
const char* escx_version(void);
/* ESC.__init__ (codecs.py:11-28, base.py:12-27,49-71): validates the configuration, derives geometry. */
int escx_create(const escx_config* cfg, int device, escx_handle* out);
void escx_destroy(escx_handle h);

/* nn.Module.load_state_dict (compress.py:23-25): one call per state_dict entry, HOST fp32, C-contiguous.
 * Keys: SURVEY.md appendix C.  `*.relative_position_index` and `*.window` are accepted and ignored
 * (the index is regenerated, the window is folded into the DFT matrices). */
int escx_set_param(escx_handle h, const char* key, const float* host_data, const int64_t* shape, int ndim);
/* Packs (pads, permutes, pre-normalises codebooks) and uploads; fails if a required key is missing. */
int escx_finalize_params(escx_handle h);
/* Number of keys escx_finalize_params() requires, and the i-th one (for load_state_dict(strict=True)). */
int escx_num_required_keys(escx_handle h);
const char* escx_required_key(escx_handle h, int i);

/* Allocates the workspace for batches up to `batch` clips of `n_samples` samples (synchronous; never
 * call while capturing a graph).  Whole-path calls reserve on demand. */
int escx_reserve(escx_handle h, int batch, int n_samples);
int64_t escx_workspace_bytes(escx_handle h);

/* ---- arithmetic of the code-emitting path ------------------------------------------------------
 * The reference computes every contraction in fp32 on ATen sgemm / bmm (esc/modules/transformer/attention.py:215-272, esc/modules/transformer/scale.py:42-145,
 * esc/modules/vq/codebook.py:31-40).  This library accumulates in fp32 everywhere; what can be chosen PER HANDLE is the form of the operands of the dense contractions
 * with K = C - fc1 / fc2 of every FeedForward, the Q / K / V projections (and, in the two-term mode, the output projection) of every WindowAttention, PatchMerge /
 * PatchSplit and, in the two-term mode, the composed PatchDeEmbed convolution (the attention's scores and P.V, the codebook search of Codebook.quantize_to_code,
 * PatchEmbed and the STFT / ISTFT GEMMs run on the fp32 MFMA in every mode):
 *   ESCX_PRECISION_FP32    every contraction on v_mfma_f32_16x16x4_f32 with fp32 operands.
 *   ESCX_PRECISION_BF16X3  every fp32 operand split EXACTLY into three bf16 terms (a = a1 + a2 + a3; bf16 has fp32's exponent range, so there is no range condition);
 *                          the six leading cross products are accumulated in fp32 on v_mfma_f32_16x16x32_bf16, smallest first.  Dropped terms < 2^-24 of a product:
 *                          the result differs from ESCX_PRECISION_FP32 only by fp32 summation order.
 *   ESCX_PRECISION_F16X2   (default) every fp32 operand a ~ a1 + a2, two fp16 terms (|a - a1 - a2| <= 2^-22 |a|), three cross products on v_mfma_f32_16x16x32_f16.
 *                          NOT exact: truncation ~1e-7 of the result magnitude, below fp32 accumulation rounding (measured: tests/test_gpu_parity.py
 *                          test_layer_accuracy_against_fp64).  Range-safe by construction: every operand is multiplied by a power of two chosen from a bound that
 *                          holds for ANY finite input (weights: max |w|; LayerNorm outputs: max |gamma| sqrt(C) + max |beta|; GELU outputs: Cauchy-Schwarz on fc1;
 *                          attention outputs: the V bound; de-embedding input: the tile's own max) and the accumulator by its inverse - exact operations - so no finite input overflows fp16 and
 *                          low terms stay normal numbers (csrc/split_terms.h).  No silent NaN, no fallback, no host synchronisation.
 * The mode is a property of the handle: it takes effect with the next call (the split weight images are re-derived on that call's stream) and a clip's codes do not
 * depend on the batch or shard it is processed in under any mode.  New handles start in the mode named by the environment variable ESCX_PRECISION (fp32 | bf16x3 | f16x2)
 * if set, else ESCX_PRECISION_F16X2.  escx_set_precision returns ESCX_ERR_INVALID_ARG for another value; escx_get_precision returns the mode in effect. */
#define ESCX_PRECISION_FP32   0
#define ESCX_PRECISION_F16X2  2
#define ESCX_PRECISION_BF16X3 3
int escx_set_precision(escx_handle h, int mode);
int escx_get_precision(escx_handle h);

/* ---- whole path ----------------------------------------------------------------------------- */
/* ESC.encode (codecs.py:68-81): wave (B,L) f32 -> codes (B,num_streams,group_size,W/overlap) int64,
 * feat_shape (H_bottom, W) written to host ints. */
int escx_encode(escx_handle h, const float* wave_dev, int batch, int n_samples, int num_streams,
                int64_t* codes_dev, int* feat_h, int* feat_w, void* stream);
/* ESC.decode (codecs.py:83-94): codes (B,S,G,T) int64 + feat_shape -> wave (B, hop*(2W-1)) f32.
 * recon_feat_dev (optional, may be NULL): (B, 2W, 2, in_freq) f32 frame-major spectrum, i.e. the
 * reference's recon_feat (B,2,F,T) permuted (0,3,1,2). */
/* Code indices outside [0, codebook_size) are clamped (the reference's F.embedding raises a device-side assert there). */
int escx_decode(escx_handle h, const int64_t* codes_dev, int batch, int num_streams, int feat_h, int feat_w,
                float* wave_out_dev, float* recon_feat_dev, void* stream);
/* ESC.forward in eval mode (codecs.py:30-66, csrvq.py:97-129): encoder once, quantise + decode in one pass.
 * raw_feat_dev (optional): (B, T, 2, in_freq) frame-major; cm_loss_dev (optional): (B,) f32 (== cb_loss). */
int escx_forward(escx_handle h, const float* wave_dev, int batch, int n_samples, int num_streams,
                 int64_t* codes_dev, float* wave_out_dev, float* raw_feat_dev, float* recon_feat_dev,
                 float* cm_loss_dev, void* stream);
/* The same with a precomputed spectrum instead of the waveform (forward(x, x_feat=...), codecs.py:33-34): feat_dev is
 * (B, T, 2, in_freq) f32 frame-major, i.e. the reference's x_feat (B,F,T,2) permuted (0,2,3,1); the STFT is skipped. */
int escx_forward_feat(escx_handle h, const float* feat_dev, int batch, int n_frames, int num_streams,
                      int64_t* codes_dev, float* wave_out_dev, float* recon_feat_dev, float* cm_loss_dev, void* stream);
int escx_num_frames(escx_handle h, int n_samples);      /* T = 1 + n_samples / hop                      */
int escx_output_samples(escx_handle h, int feat_w);     /* hop * (patch_t * W - 1)                      */

/* ---- stage level (reference layouts; used by the parity tests) ------------------------------- */
/* BaseAudioCodec.spec_transform (base.py:29-37): (B,L) -> (B,T,2,F) frame-major [= (B,2,F,T).permute(0,3,1,2)] */
int escx_spec_transform(escx_handle h, const float* wave_dev, int batch, int n_samples, float* spec_dev, void* stream);
/* BaseAudioCodec.audio_reconstruct (base.py:39-47): (B,T,2,F) -> (B, hop*(T-1)) */
int escx_audio_reconstruct(escx_handle h, const float* spec_dev, int batch, int n_frames, float* wave_dev, void* stream);
/* PatchEmbed.forward (scale.py:42-50): spec (B,T,2,F) -> tokens (B, H*W, C0) */
int escx_patch_embed(escx_handle h, const float* spec_dev, int batch, int n_frames, float* tokens_dev, void* stream);
/* TransformerLayer.forward (attention.py:48-91).  layer_id: 0 = encoder.pre_nn, 1..n-1 = encoder.blocks[i-1],
 * n..2n-2 = decoder.blocks[i-n], 2n-1 = decoder.post_nn  (n = n_scales).  x (B,H*W,C) -> y (B,H'*W,C'). */
int escx_transformer_layer(escx_handle h, int layer_id, const float* x_dev, int batch, int H, int W,
                           float* y_dev, int* H_out, void* stream);
/* CrossScaleRVQ.csrvq_encode / PVQ.encode (csrvq.py:50-54, quantization.py:74-91, codebook.py:20-43):
 * codes[b,g,t] for residual = enc - dec (dec may be NULL -> residual = enc).  codes_dev: (B,G,T) int64
 * with `code_batch_stride` elements between clips (lets the caller write straight into (B,S,G,T)). */
int escx_pvq_encode(escx_handle h, int stream_id, const float* enc_dev, const float* dec_dev, int batch, int W,
                    int64_t* codes_dev, int64_t code_batch_stride, void* stream);
/* CrossScaleRVQ.csrvq_decode / PVQ.decode (csrvq.py:56-60, quantization.py:93-108): out = dec + dequant(codes). */
int escx_pvq_decode(escx_handle h, int stream_id, const int64_t* codes_dev, int64_t code_batch_stride,
                    const float* dec_dev, int batch, int W, float* out_dev, void* stream);
/* PatchDeEmbed.forward (scale.py:73-81): tokens (B,H0*W,C0) -> spec (B, 2W, 2, F) frame-major */
int escx_patch_deembed(escx_handle h, const float* tokens_dev, int batch, int W, float* spec_dev, void* stream);

/* ---- per-kernel timing (HIP events recorded on the caller's stream around every launch) ------- */
/* enable != 0 starts recording (and clears earlier records); enable == 0 stops.  enable == 2 additionally runs the batch parts
 * back to back on the caller's stream, so that each kernel is timed alone on the GPU. */
int escx_profile_enable(escx_handle h, int enable);
/* Synchronises, aggregates by kernel label and returns a JSON array
 * [{"name":..., "calls":n, "ms":total, "flops":algorithmic, "bytes":algorithmic}, ...] valid until the next call. */
const char* escx_profile_report(escx_handle h);

/* Device math used inside the fused kernels, exposed for the accuracy tests: which = 0 gelu (branch-free erf),
 * 1 erf (branch-free), 2 exp via v_exp_f32, 3 gelu via libm erff (the unfused epilogue). */
/* Debug: per-wave cycle counters of the fused MLP main loop (kernel built with the trace flag, ESCX_MLP_VARIANT=164). */
int escx_debug_mlp_trace(unsigned long long* dev_buf);
int escx_test_math(const float* x_dev, float* y_dev, int64_t n, int which, void* stream);
/* Counter calibration (tools/pmc_calib.py): copies `rows` rows of `row_floats` floats with the fused kernels' access pattern (16 lanes per
 * row, 16 B per lane), so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be compared with a known byte count. */
int escx_test_copy_rows(const float* src_dev, float* dst_dev, int64_t rows, int row_floats, void* stream);
/* Host-side evaluation of the multiply-high division the gather loaders / scatter epilogues use on the device (gemm_engine.h FastDiv):
 * returns n / d for 0 <= n < 2^31, 1 <= d < 2^31 (no GPU needed; tests compare it with exact division). */
int escx_test_fastdiv(int n, int d);

/* ---- code packing for transport (10-bit codes; all-gather payload) --------------------------- */
/* 10-bit wire format (codebook_size 1024): n codes <-> 5*ceil(n/4) bytes; 6 streams x 3 groups x 50 Hz x 10 b = 9 kbps (base.py:70). */
int escx_codes_pack10(const int64_t* codes_dev, uint8_t* out_dev, int64_t n, void* stream);
int escx_codes_unpack10(const uint8_t* in_dev, int64_t* codes_dev, int64_t n, void* stream);
int escx_codes_narrow(const int64_t* codes_dev, int16_t* out_dev, int64_t n, void* stream);
int escx_codes_widen(const int16_t* in_dev, int64_t* codes_dev, int64_t n, void* stream);

/* ---- training step (BASELINE configs[4]) --------------------------------------------------------------
 * Reference: scripts/trainer_no_adv.py:95-118 (step), esc/models/codecs.py:30-66 + esc/models/csrvq.py:23-48,97-129 (training-mode
 * forward), esc/modules/vq/codebook.py:57-75 (straight-through estimator, commitment / codebook losses),
 * esc/modules/vq/quantization.py:53-64 (freeze_vq), esc/modules/loss/generator_loss.py:12-74 (losses).  fp32, like the reference.
 * Trainable parameters live in ONE flat fp32 device buffer owned by the caller, in the order escx_flat_param_*() reports (the keys
 * escx_finalize_params requires, reference shapes, C-contiguous); gradients are returned in a buffer of the same layout. */
int escx_flat_param_count(escx_handle h);
const char* escx_flat_param_key(escx_handle h, int i);
int64_t escx_flat_param_offset(escx_handle h, int i);        /* in floats */
int64_t escx_flat_param_numel(escx_handle h, int i);
int64_t escx_flat_param_total(escx_handle h);
/* Re-derives every packed layout from the flat buffer.  full == 0: on the device (gather + codebook normalisation; asynchronous),
 * enough for the training step; full != 0: additionally rebuilds the host-folded layouts of the inference path (synchronous). */
int escx_load_flat_params(escx_handle h, const float* flat_params_dev, int full, void* stream);
/* ESC.forward in training mode.  flat_params_dev may be NULL (keep the currently packed weights).  codes_dev: (B, max_streams, G, T)
 * int64 - every stream is quantised in training mode (csrvq.py:104-113); wave_out (B, hop*(2W-1)); raw_feat (B,T,2,F) and recon_feat
 * (B,2W,2,F) frame-major, optional; cm_loss / cb_loss (B,), optional.  Activations stay on the handle's tape until the backward. */
int escx_train_forward(escx_handle h, const float* flat_params_dev, const float* wave_dev, int batch, int n_samples, int num_streams,
                       int freeze_codebook, int64_t* codes_dev, float* wave_out_dev, float* raw_feat_dev, float* recon_feat_dev,
                       float* cm_loss_dev, float* cb_loss_dev, void* stream);
/* The same with a precomputed spectrum instead of the waveform (forward(x, x_feat=...) in training mode, codecs.py:33-34): feat_dev is (B, T, in_dim, F) f32 frame-major,
 * i.e. the reference's x_feat (B,F,T,2) permuted (0,2,3,1); the STFT is skipped, everything downstream (tape, backward) is escx_train_forward's. */
int escx_train_forward_feat(escx_handle h, const float* flat_params_dev, const float* feat_dev, int batch, int n_frames, int num_streams, int freeze_codebook,
                            int64_t* codes_dev, float* wave_out_dev, float* recon_feat_dev, float* cm_loss_dev, float* cb_loss_dev, void* stream);
/* Backward of the last escx_train_forward.  Upstream gradients (any may be NULL = zero): d_wave (B, out_len), d_recon_feat (B,2W,2,F),
 * d_cm_loss / d_cb_loss (B,).  grad_flat_dev receives d loss / d parameter in the flat layout (overwritten, not accumulated). */
int escx_train_backward(escx_handle h, const float* d_wave_dev, const float* d_recon_feat_dev, const float* d_cm_loss_dev,
                        const float* d_cb_loss_dev, float* grad_flat_dev, void* stream);
int64_t escx_train_tape_bytes(escx_handle h);
/* Number of escx_train_forward calls on this handle so far.  The handle holds ONE tape: escx_train_backward consumes the activations of the
 * LAST forward.  A caller that interleaves several forwards (autograd graphs alive at the same time) records this value after its forward
 * and compares it before its backward; a mismatch means the tape belongs to a later forward (esc/models/codecs.py raises). */
int64_t escx_train_tape_generation(escx_handle h);
/* ComplexSTFTLoss (generator_loss.py:12-35): per-clip loss (B,) = mean over the clip's `per_clip` spectrum values (any layout, the
 * same for both inputs) of (pl(raw) - pl(recon))^2, pl = power-law compression; optionally d loss_b / d recon_feat. */
int escx_stft_loss(const float* raw_feat_dev, const float* recon_feat_dev, int batch, int64_t per_clip, float* loss_dev,
                   float* d_recon_feat_dev, void* stream);
/* MelSpectrogramLoss (generator_loss.py:37-74; 7 resolutions, windows 32..2048, hop = window/4, HTK mel filterbanks of 5..320 bins,
 * L1 on the mel magnitudes + L1 on log10(clamp(mel)^2)): per-clip loss (B,) and, optionally, d loss_b / d recon_wave (B,L).
 * Runs on the current device; the DFT / filterbank matrices are built once per device. */
int escx_mel_loss(const float* raw_wave_dev, const float* recon_wave_dev, int batch, int n_samples, int sample_rate, float* loss_dev,
                  float* d_recon_wave_dev, void* stream);
int escx_scale_rows(const float* x_dev, const float* g_dev, float* out_dev, int rows, int64_t per_row, void* stream);   /* out[r][:] = x[r][:] * g[r] */
/* clip_grad_norm_ + AdamW on flat buffers (trainer_no_adv.py:116-117).  norm_out_dev: 2 + 1024 floats ([0] = norm, [1] = clip coefficient). */
int escx_grad_norm_clip(const float* grad_flat_dev, int64_t n, float max_norm, float* norm_out_dev, void* stream);
/* Hyper-parameters are doubles, as torch.optim.AdamW holds them (python floats): 1 - beta, lr / (1 - beta1^t), sqrt(1 - beta2^t) and 1 - lr * wd are
 * evaluated in double on the host and reach the kernel as fp32 scalars, exactly as torch's single-tensor path does. */
int escx_adamw_step(float* param_flat_dev, const float* grad_flat_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n, int step,
                    double lr, double beta1, double beta2, double eps, double weight_decay, const float* clip_dev, void* stream);

/* ---- adversarial step: DAC discriminator + GAN losses (BASELINE configs[4]; scripts/trainer_adv.py:61-107) ---------------------------------
 * Reference: esc/models/discriminator.py:31-221 (MPD :31-66, MRD :105-176, Discriminator :179-215; MSD is not used by any ESC config),
 * esc/modules/loss/gan_loss.py:5-51.  Parameters: one flat fp32 device buffer in the order escx_disc_param_*() reports (the reference's
 * named_parameters(): per convolution bias, weight_g, weight_v).  Feature maps are caller-owned channels-last buffers [B][D0][P1][Cp]
 * (escx_disc_fmap_shape); the reference's (B, C, D0, D1) tensor of map i is buffer[:, :, off1 : off1 + D1, :C] permuted (0, 3, 1, 2). */
typedef struct {
    int32_t sample_rate;                 /* 16000                                                  */
    int32_t n_rates;                     /* must be 0 (MSD unsupported)                            */
    int32_t n_periods; int32_t periods[8];      /* 2, 3, 5, 7, 11                                  */
    int32_t n_ffts; int32_t fft_sizes[8];       /* 2048, 1024, 512                                 */
    int32_t n_bands; float bands[8][2];         /* (0, .1), (.1, .25), (.25, .5), (.5, .75), (.75, 1) */
} escx_disc_config;
typedef struct escx_disc_s* escx_disc;
int escx_disc_create(const escx_disc_config* cfg, int device, escx_disc* out);
void escx_disc_destroy(escx_disc d);
int escx_disc_param_count(escx_disc d);
const char* escx_disc_param_key(escx_disc d, int i);
int64_t escx_disc_param_offset(escx_disc d, int i);
int64_t escx_disc_param_numel(escx_disc d, int i);
int64_t escx_disc_param_total(escx_disc d);
int escx_disc_num_fmaps(escx_disc d, int n_samples);
int escx_disc_fmap_shape(escx_disc d, int n_samples, int i, int* sub, int* C, int* Cp, int* D0, int* D1, int* P1, int* off1);
/* Discriminator.forward: wave (B, L) -> every feature map (fmaps_dev: HOST array of device pointers, map i's own base address).
 * params_version: a number that changes whenever the CONTENTS of the flat buffer change (negative: unknown); the weight-normalised GEMM operands
 * are re-derived from the flat buffer only when (pointer, version) differs from the previous call. */
int escx_disc_forward(escx_disc d, const float* flat_params_dev, int64_t params_version, const float* wave_dev, int batch, int n_samples,
                      float* const* fmaps_dev, void* stream);
/* Backward of a forward on the same (params, wave, fmaps): d_fmaps_dev[i] (same layout, NULL = zero) -> grad_flat_dev (optional, overwritten)
 * and / or d_wave_dev (optional, (B, L), overwritten). */
int escx_disc_backward(escx_disc d, const float* flat_params_dev, int64_t params_version, const float* wave_dev, int batch, int n_samples,
                       float* const* fmaps_dev, const float* const* d_fmaps_dev, float* grad_flat_dev, float* d_wave_dev, void* stream);
/* Arithmetic of the discriminator's wide convolutions (round 4; BASELINE configs[4] names bf16, the reference's trainer_adv.py runs fp32).
 * 0 (default): fp32 MFMA everywhere, the parity-tested path.  1: the implicit GEMMs with at least 128 output and 256 contraction columns (the 128 -> 512 ->
 * 1024 -> 1024 period convolutions: forward, dX, dW) and the 32 -> 32-channel band convolutions of the spectrogram discriminators (forward, dX, dW) round
 * their operands to bf16 (nearest even) while staging them and accumulate in fp32 on the bf16 MFMA;
 * feature maps, parameters, gradients and every other kernel stay fp32.
 * 2 (round 5; the default of the Python host's Discriminator): the same wide implicit GEMMs (forward, dX, dW of the period convolutions) on the bf16 MFMA with BOTH
 * operands split exactly into three bf16 terms (a = a1 + a2 + a3) and the six leading cross products accumulated in fp32, smallest first - fp32-grade results
 * (within fp32 summation-order noise of mode 0; the truncated terms are below 2^-26 relative); the band convolutions stay on the fp32 MFMA.
 * Returns ESCX_ERR_INVALID_ARG for another mode. */
int escx_disc_set_precision(escx_disc d, int mode);
int escx_disc_get_precision(escx_disc d);
/* One GAN loss term over a feature-map buffer (gan_loss.py:30-51): loss_dev[b] (+)= mean over the C x D0 x D1 real elements of
 * (target - x)^2 (mode 0) or |x - ref| (mode 1); grad_dev (optional, layout of x) receives d term_b / d x. */
int escx_gan_term(const float* x_dev, const float* ref_dev, float* grad_dev, int batch, int C, int Cp, int D0, int D1, int P1, int mode, float target,
                  float* loss_dev, int accumulate, void* stream);
/* Backward of escx_gan_term with the upstream per-clip gradient folded in (what autograd does with `loss.mean().backward()` in trainer_adv.py:77-105):
 * grad_dev (layout of x) = g_dev[b] * d term_b / d x.  The forward keeps no per-map gradient buffer; x (and ref) are re-read here. */
int escx_gan_term_grad(const float* x_dev, const float* ref_dev, const float* g_dev, float* grad_dev, int batch, int C, int Cp, int D0, int D1, int P1, int mode,
                       float target, void* stream);
/* Per-layer timing of the discriminator's convolutions (diagnostic: HIP events and a stream sync around every launch group while enabled, so the step
 * itself runs slower).  escx_disc_profile_enable(1) clears and starts, (0) stops; the report is a JSON array of {"name": "D.<fwd|dX|dW>[<layer>]",
 * "calls", "ms", "flops"} - the same record format as escx_profile_report.  Process-wide (not per handle). */
int escx_disc_profile_enable(int enable);
const char* escx_disc_profile_report(void);

/* ---- multi-GPU: the one exchange step of the sharded path (BASELINE configs[3]; SURVEY.md 8(b),(e)) ------------- */
/* The reference has no inference-time collective (clips are independent end to end); batch shards exchange only the emitted codes.
 * codes_local_dev: n_local_codes int64 values of this rank (e.g. 36*S*G*T); codes_all_dev: world_size * n_local_codes, rank order.
 * The codes cross xGMI as int16 (RCCL byte payload).  nccl_comm is the caller's ncclComm_t; libescx resolves RCCL at run time
 * (dlopen of "librccl.so.1": in a PyTorch process torch's bundled copy, already mapped) so that communicator and collective come
 * from one RCCL instance; escx_set_rccl_library(path) overrides the name before the first call.  Asynchronous on `stream`, except
 * that growing the int16 staging buffer synchronises (first call / larger shard).  Equal shard sizes on every rank. */
int escx_set_rccl_library(const char* path);
int escx_allgather_codes(escx_handle h, const int64_t* codes_local_dev, int64_t n_local_codes, int64_t* codes_all_dev,
                         int world_size, void* nccl_comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESCX_H */
