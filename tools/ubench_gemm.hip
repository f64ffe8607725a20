// Micro-benchmark of the fp32 MFMA GEMM engine (csrc/gemm_engine.h) on the shapes that dominate the training / adversarial step:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I efficient-speech-codec_amd/csrc tools/ubench_gemm.hip -o gpurun_out/ubench_gemm && gpurun_out/ubench_gemm
// Prints TFLOP/s per (kernel variant, shape).  Tuning aid only: not part of libescx.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "gemm_engine.h"
#include "gemm_engine2.h"
#define HAVE_V2 1

using namespace escx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

struct Shape { const char* name; int M, N, K; };

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;          // shape index, -1 = all
    int si = -1;
    const Shape shapes[] = {{"disc 1024x5120 (MPD conv5 as plain GEMM)", 21504, 1024, 5120}, {"steady state 1024x2048, M=172032", 172032, 1024, 2048}, {"disc 1024x2560", 21504, 1024, 2560},
                            {"fc1 C=384 (M=21600,N=1536,K=384)", 21600, 1536, 384}, {"fc1 C=96 (M=172800,N=384,K=96)", 172800, 384, 96},
                            {"fc1 C=45 (M=691200,N=192,K=48)", 691200, 192, 48}, {"fc2 C=45 (M=691200,N=48,K=192)", 691200, 48, 192}};
    size_t maxA = 0, maxW = 0, maxO = 0;
    for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxO = std::max(maxO, (size_t)s.M * s.N); }
    float *A, *W, *O;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&W, maxW * 4)); CK(hipMalloc(&O, maxO * 4));
    std::vector<float> h(std::max(maxA, maxW));
    unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), maxW * 4, hipMemcpyHostToDevice));
    for (auto& sh : shapes) {
        ++si; if (only >= 0 && si != only) continue;
        const double fl = 2.0 * sh.M * sh.N * sh.K;
        const double by = 4.0 * ((double)sh.M * sh.K + (double)sh.M * sh.N);
        printf("== %s: %.1f GFLOP, %.0f MB in+out\n", sh.name, fl / 1e9, by / 1e6);
        PlainA ld{A, sh.K, sh.M}; EpiStore ep{O, sh.N, nullptr};
        auto run = [&](const char* tag, auto fn) {
            const float ms = time_ms(fn, 8);
            printf("   %-34s %8.3f ms  %7.1f TFLOP/s  %7.1f GB/s\n", tag, ms, fl / ms / 1e9, by / ms / 1e6);
        };
        run("engine<128> default", [&] { launch_gemm<128>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("engine<128> bk16", [&] { launch_gemm<128>(ld, W, sh.M, sh.N, sh.K, ep, 0, 1, 16); });
        if (sh.K % 32 == 0) run("engine<128> bk32", [&] { launch_gemm<128>(ld, W, sh.M, sh.N, sh.K, ep, 0, 1, 32); });
        if (sh.K % 48 == 0) run("engine<128> bk48", [&] { launch_gemm<128>(ld, W, sh.M, sh.N, sh.K, ep, 0, 1, 48); });
        run("engine<64> bk16", [&] { launch_gemm<64>(ld, W, sh.M, sh.N, sh.K, ep, 0, 1, 16); });
        // round 5: the same loop on v_mfma_f32_32x32x2_f32 (gemm_engine2.h gemm_kernel_m32); results agree to rounding, not bitwise (other k order)
        run("m32<128,96,16>  (32x32x2 MFMA)", [&] { launch_gemm_m32<96, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("m32<128,128,16> (32x32x2 MFMA)", [&] { launch_gemm_m32<128, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        if (sh.K % 32 == 0) run("m32<128,128,32> (32x32x2 MFMA)", [&] { launch_gemm_m32<128, 32>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("m32<128,64,16>  (32x32x2 MFMA)", [&] { launch_gemm_m32<64, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        {   // agreement with the engine (relative error of 4096 probed outputs)
            launch_gemm<128>(ld, W, sh.M, sh.N, sh.K, ep, 0, 1, 16); hipDeviceSynchronize();
            std::vector<float> r0(4096), r1(4096);
            hipMemcpy(r0.data(), O + (size_t)(sh.M - 70) * sh.N, 4096 * 4, hipMemcpyDeviceToHost);
            hipMemset(O, 0, (size_t)sh.M * sh.N * 4);
            launch_gemm_m32<128, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); hipDeviceSynchronize();
            hipMemcpy(r1.data(), O + (size_t)(sh.M - 70) * sh.N, 4096 * 4, hipMemcpyDeviceToHost);
            double num = 0, den = 0; for (int i = 0; i < 4096; ++i) { num += (double)(r0[i] - r1[i]) * (r0[i] - r1[i]); den += (double)r0[i] * r0[i]; }
            printf("   m32 vs engine: relative rms difference %.2e over 4096 probed outputs\n", std::sqrt(num / (den > 0 ? den : 1)));
        }
#ifdef HAVE_V2
        run("v3<128,96>", [&] { launch_gemm3<128, 96>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("v3<128,128>", [&] { launch_gemm3<128, 128>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("v3<256,96>", [&] { launch_gemm3<256, 96>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("v3<256,128>", [&] { launch_gemm3<256, 128>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        {   // same results as the engine?
            launch_gemm<128>(ld, W, sh.M, sh.N, sh.K, ep, 0, 1, 16); hipDeviceSynchronize();
            std::vector<float> r0(4096), r1(4096);
            hipMemcpy(r0.data(), O + (size_t)(sh.M - 70) * sh.N, 4096 * 4, hipMemcpyDeviceToHost);
            hipMemset(O, 0, (size_t)sh.M * sh.N * 4);
            launch_gemm3<128, 96>(ld, W, sh.M, sh.N, sh.K, ep, 0); hipDeviceSynchronize();
            hipMemcpy(r1.data(), O + (size_t)(sh.M - 70) * sh.N, 4096 * 4, hipMemcpyDeviceToHost);
            int bad = 0; for (int i = 0; i < 4096; ++i) bad += (r0[i] != r1[i]);
            printf("   v3 vs engine: %d of 4096 probed outputs differ bitwise\n", bad);
        }
        run("v2<128,96,16>", [&] { launch_gemm2<128, 96, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        if (sh.K % 32 == 0) run("v2<128,96,32>", [&] { launch_gemm2<128, 96, 32>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("v2<128,128,16>", [&] { launch_gemm2<128, 128, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        if (sh.K % 32 == 0) run("v2<128,128,32>", [&] { launch_gemm2<128, 128, 32>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("v2<256,96,16>", [&] { launch_gemm2<256, 96, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
        run("v2<256,128,16>", [&] { launch_gemm2<256, 128, 16>(ld, W, sh.M, sh.N, sh.K, ep, 0); });
#endif
    }
    return 0;
}
