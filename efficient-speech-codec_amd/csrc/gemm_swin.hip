// GEMM instantiations for the Swin block and the scale-change linears.
#include "gemm_engine.h"
#include "launchers.h"

namespace escx {

template <class Epi>
static void plain_gemm(const float* A, int lda, int M, const float* W, int Np, int Kp, const Epi& ep, hipStream_t s) {
    PlainA ld{A, lda, M};
    // 128-row tiles halve the weight traffic per output row; keep 64 when the grid would not fill 256 CUs
    const long long tiles128 = (long long)((M + 127) / 128) * ((Np + 95) / 96);
    if (tiles128 >= 512) launch_gemm<128>(ld, W, M, Np, Kp, ep, s);
    else launch_gemm<64>(ld, W, M, Np, Kp, ep, s);
}

void gemm_qkv(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* bias, int nq, float scale,
              hipStream_t s) {
    plain_gemm(A, lda, M, W, Np, Kp, EpiQkv{out, Np, bias, nq, scale}, s);
}
void gemm_proj_scatter(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* shortcut,
                       const float* bias, const int* map, int slots, int tokens, hipStream_t s) {
    plain_gemm(A, lda, M, W, Np, Kp, EpiProjScatter{out, shortcut, bias, map, slots, tokens, Np}, s);
}
void gemm_gelu(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* bias, hipStream_t s) {
    plain_gemm(A, lda, M, W, Np, Kp, EpiGelu{out, Np, bias}, s);
}
void gemm_residual(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, const float* bias, const float* res,
                   hipStream_t s) {
    plain_gemm(A, lda, M, W, Np, Kp, EpiResidual{out, Np, bias, res}, s);
}
void gemm_store(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, int ldo, const float* bias, hipStream_t s) {
    plain_gemm(A, lda, M, W, Np, Kp, EpiStore{out, ldo, bias}, s);
}
void gemm_split(const float* A, int lda, int M, const float* W, int Np, int Kp, float* out, int H, int Wd, int C2p, hipStream_t s) {
    plain_gemm(A, lda, M, W, Np, Kp, EpiSplit{out, H, Wd, C2p}, s);
}

}  // namespace escx
