// GEMM instantiations with gathering A-side loaders: STFT framing, patch embedding, de-embedding
// convolutions (implicit GEMM), product-VQ down / up projections.
#include "gemm_engine.h"
#include "launchers.h"

namespace escx {

void gemm_frames(const float* wave, int B, int L, int T, int hop, int off, const float* W, int Np, int Kp, float* out, hipStream_t s) {
    FrameA ld{wave, L, T, hop, off, B * T, FastDiv(T)};
    launch_gemm<64>(ld, W, B * T, Np, Kp, EpiStore{out, Np, nullptr}, s);
}

void gemm_patch(const float* spec, int B, int T, int in_dim, int Fp, int H, int Wd, int pf, int pt, const float* W, int Np, int Kp,
                float* out, const float* bias, hipStream_t s) {
    PatchA ld{spec, T, in_dim * Fp, Fp, H, Wd, pf, pt, in_dim * pf * pt, B * H * Wd};
    launch_gemm<64>(ld, W, B * H * Wd, Np, Kp, EpiStore{out, Np, bias}, s);
}

void gemm_conv_deembed1(const float* x, int B, int H, int Wd, int Cp, const float* W, int Np, float* out, const float* bias, int pf,
                        int pt, hipStream_t s) {
    ConvA ld{x, H, Wd, Cp, 5, 5, B * H * Wd};
    const int M = B * H * Wd;
    if (M >= 32768) launch_gemm<128>(ld, W, M, Np, 25 * Cp, EpiDeembed1{out, bias, H, Wd, Cp, pf, pt}, s, 1, pick_bk(Cp));
    else launch_gemm<64>(ld, W, M, Np, 25 * Cp, EpiDeembed1{out, bias, H, Wd, Cp, pf, pt}, s, 1, pick_bk(Cp));
}

void gemm_conv_spec(const float* x, int B, int T, int F, int Cp, const float* W, float* out, const float* bias, int Fp, int in_dim,
                    hipStream_t s) {
    ConvA ld{x, T, F, Cp, 3, 3, B * T * F};
    launch_gemm<64>(ld, W, B * T * F, 16, 9 * Cp, EpiSpec{out, bias, T, F, Fp, in_dim}, s, 1, pick_bk(Cp));
}

int test_fastdiv(int n, int d) { return FastDiv(d).div(n); }

void gemm_deembed_composed(const float* x, int B, int H, int Wd, int Cp, const float* W, float* out, const float* bias, int pf, int pt,
                           int in_dim, int Fp, hipStream_t s) {
    ConvA ld{x, H, Wd, Cp, 7, 7, B * H * Wd};
    const int M = B * H * Wd;
    EpiDeembedC ep{out, bias, H, Wd, pf, pt, in_dim, Fp};
    if (M >= 32768) launch_gemm<128>(ld, W, M, 16, 49 * Cp, ep, s, 1, pick_bk(Cp));
    else launch_gemm<64>(ld, W, M, 16, 49 * Cp, ep, s, 1, pick_bk(Cp));
}

// Split-K factor of the PVQ down-projection.  It must not depend on the batch: the partial sums are added in a fixed order, but a
// different number of slices would re-associate the fp32 sum and could move a near-tie code, i.e. a clip would no longer get the
// same codes in every batch / shard it is processed in (test_node_batch_288_matches_its_36_clip_shards).
int pvq_down_splits(int /*M*/, int Kp, int Cp) {
    const int BK = pick_bk(Cp);
    const int kIters = Kp / BK;
    int splits = kIters / 2;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
    const int per = (kIters + splits - 1) / splits;
    return (kIters + per - 1) / per;
}

int pvq_down_bk(int Cp) { return pick_bk(Cp); }

void gemm_pvq_down(const float* enc, const float* dec, int B, int Hq, int Wd, int Cp, int ov, const float* W, int Np, int Kp,
                   float* zpart, int splits, hipStream_t s) {
    const int Tq = Wd / ov, M = B * Tq;
    ResidualGatherA ld{enc, dec, Hq, Wd, Cp, Tq, ov, M, FastDiv(Tq), FastDiv(Cp), FastDiv(Hq)};
    launch_gemm<64>(ld, W, M, Np, Kp, EpiPartial{zpart, M, Np}, s, splits, pick_bk(Cp));
}

void gemm_pvq_up(const long long* codes, long long bstride, const float* cbraw, int G, int Ksz, int dt, int B, int Hq, int Wd, int Cp,
                 int ov, const float* W, int Np, int Kp, const float* dec, float* out, hipStream_t s) {
    const int Tq = Wd / ov, M = B * Tq;
    CodeGatherA ld{codes, bstride, cbraw, G, Ksz, dt, Tq, M, FastDiv(Tq), FastDiv(dt)};
    launch_gemm<64>(ld, W, M, Np, Kp, EpiPvqAdd{out, dec, Hq, Wd, Cp, Tq, ov, FastDiv(Tq), FastDiv(Cp), FastDiv(Hq)}, s);
}

}  // namespace escx
