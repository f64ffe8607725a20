// Launchers for the non-GEMM kernels.
#include "kernels.h"
#include "fused_pvq.h"
#include <atomic>
#include "launchers.h"

namespace escx {

static inline int grid_for(long long work_items, int per_block, int cap = 4096) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

void ln_rows(int mode, const float* src, float* dst, const float* gamma, const float* beta, const int* map, int rows_per_clip,
             int src_rows_per_clip, int total_rows, int C, int Cp, hipStream_t s) {
    const int grid = grid_for(total_rows, 16, 8192);
    const float eps = 1e-5f;
    if (mode == 0)
        ESCX_LAUNCH((ln_rows_kernel<1, 0>), dim3(grid), dim3(256), 0, s, src, dst, gamma, beta, map, rows_per_clip,
                           src_rows_per_clip, total_rows, C, Cp, eps);
    else if (mode == 1)
        ESCX_LAUNCH((ln_rows_kernel<1, 1>), dim3(grid), dim3(256), 0, s, src, dst, gamma, beta, map, rows_per_clip,
                           src_rows_per_clip, total_rows, C, Cp, eps);
    else
        ESCX_LAUNCH((ln_rows_kernel<2, 2>), dim3(grid), dim3(256), 0, s, src, dst, gamma, beta, map, rows_per_clip,
                           src_rows_per_clip, total_rows, C, Cp, eps);
}

int window_attention(const float* qkv, const float* bias, float* out, int total_windows, int nH, int hdp, int ldq, int ldo, int nWh,
                     int nWw, int shifted, hipStream_t s) {
    const long long pairs = (long long)total_windows * nH;
    const int grid = grid_for(pairs, 4, 16384);
#define ESCX_ATT(S) case S: ESCX_LAUNCH((window_attention_kernel<S>), dim3(grid), dim3(256), 0, s, qkv, bias, out, (int)pairs, \
                                               nH, ldq, ldo, nWh, nWw, shifted); return 0;
    switch (hdp / 4) {
        ESCX_ATT(1) ESCX_ATT(2) ESCX_ATT(3) ESCX_ATT(4) ESCX_ATT(5) ESCX_ATT(6) ESCX_ATT(7) ESCX_ATT(8)
        ESCX_ATT(12) ESCX_ATT(16)
        default: return -1;
    }
#undef ESCX_ATT
}

int window_attention_any(const float* qkv, const float* bias, float* out, int total_windows, int ws, int nH, int hd, int hdp, int ldq, int ldo, int nWh, int nWw,
                         int shift, hipStream_t s) {
    if (total_windows < 1 || nH < 1 || nH > 65535 || ws < 1) return -1;
    ESCX_LAUNCH(window_attention_any_kernel, dim3((unsigned)total_windows, (unsigned)nH), dim3(64), 0, s, qkv, bias, out, nH, hd, hdp, ldq, ldo, ws, nWh, nWw, shift);
    return 0;
}

int pvq_search(const float* zpart, int splits, int M, int ldz, const float* cbn, const float* c2, const float* cbraw, int G, int Ksz,
               int d, int dt, int Tq, long long* codes, long long bstride, float* loss, float loss_scale, int l2norm, hipStream_t s) {
    SearchArgs a{zpart, splits, M, ldz, cbn, c2, cbraw, Ksz, d, Tq, codes, bstride, loss, loss_scale, l2norm};
    dim3 grid((M + 15) / 16, G);
#define ESCX_SRCH(S) case S: ESCX_LAUNCH((pvq_search_kernel<S>), grid, dim3(256), 0, s, a); return 0;
    switch (dt / 4) {
        ESCX_SRCH(1) ESCX_SRCH(2) ESCX_SRCH(3) ESCX_SRCH(4) ESCX_SRCH(5) ESCX_SRCH(6) ESCX_SRCH(7) ESCX_SRCH(8)
        ESCX_SRCH(12) ESCX_SRCH(16)
        default: return -1;
    }
#undef ESCX_SRCH
}

void loss_reduce(const float* terms, int n_slots, int G, int M, int Tq, float* out, hipStream_t s) {
    ESCX_LAUNCH(loss_reduce_kernel, dim3(M / Tq), dim3(64), 0, s, terms, n_slots, G, M, Tq, out);
}

int pvq_down(const float* enc, const float* dec, int B, int Hq, int Wd, int Cp, int ov, const float* W, int Np, int Kp, float* zpart, int splits,
             int bk, hipStream_t s) {
#ifdef ESCX_EXPERIMENTAL       // bit-identical to the engine form but 30 % slower alone (profiles/r4_pvq_ab.txt): tagged builds only
    const int Tq = Wd / ov, M = B * Tq;
    if (Np % 16 || Kp % 16 || Cp % 16 || bk % 16 || Kp % bk) return -1;
    const int kIters = Kp / bk, per = (kIters + splits - 1) / splits;      // gemm_engine.h launch_tile: the SAME slice boundaries as the engine's split-K
    if ((kIters + per - 1) / per != splits) return -1;
    PvqDownArgs a{enc, dec, W, zpart, M, Tq, Hq, Wd, Cp, ov, Kp, Np, per * bk};
    const dim3 grid((M + 63) / 64, splits);
    switch (Np / 16) {
        case 1: ESCX_LAUNCH(pvq_down_kernel<1>, grid, dim3(256), 0, s, a); return 0;
        case 2: ESCX_LAUNCH(pvq_down_kernel<2>, grid, dim3(256), 0, s, a); return 0;
        case 3: ESCX_LAUNCH(pvq_down_kernel<3>, grid, dim3(256), 0, s, a); return 0;
        case 4: ESCX_LAUNCH(pvq_down_kernel<4>, grid, dim3(256), 0, s, a); return 0;
        case 6: ESCX_LAUNCH(pvq_down_kernel<6>, grid, dim3(256), 0, s, a); return 0;
        default: return -1;
    }
#else
    return -1;
#endif
}

// One product-VQ stream in one launch (fused_pvq.h).  -1: geometry not covered, the caller runs the three-launch form.
template <int NT, int STEPS, bool DEC>
static void launch_pvq_fused_d(const PvqFusedArgs& a, hipStream_t s) {
    auto kern = pvq_fused_kernel<NT, STEPS, DEC>;
    constexpr int lds = pvqf_lds_floats<NT>() * (int)sizeof(float);
    if constexpr (lds > 48 * 1024) {            // function attributes are per device: one flag per device, set before the first launch there
        static std::atomic<unsigned> done{0};
        int dev = 0; (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    ESCX_LAUNCH(kern, dim3((a.M + 15) / 16), dim3(64 * PVQF_WAVES), lds, s, a);
}
template <int NT, int STEPS>
static void launch_pvq_fused(const PvqFusedArgs& a, hipStream_t s) {
    if (a.dec) launch_pvq_fused_d<NT, STEPS, true>(a, s); else launch_pvq_fused_d<NT, STEPS, false>(a, s);
}

void pvq_tab_add(const long long* codes, long long bstride, const float* tab, const float* gq, int G, int Ksz, int B, int Hq, int Wd, int Cp, int ov,
                 const float* dec, float* out, hipStream_t s) {
    PvqTabAddArgs a{codes, bstride, tab, gq, dec, out, G, Ksz, Wd / ov, Hq, Wd, Cp, ov, (long long)B * Hq * Wd * (Cp / 4)};
    ESCX_LAUNCH(pvq_tab_add_kernel, dim3((unsigned)((a.n4 + 255) / 256)), dim3(256), 0, s, a);
}

int pvq_fused(const float* enc, const float* dec, int B, int Hq, int Wd, int Cp, int ov, const float* wd, int Np, int Kq, int splits, int bk,
              const float* cbn, const float* c2, const float* cbraw, int G, int Ksz, int d, int dt, const float* wup, const float* tab, const float* gq,
              float* out, long long* codes, long long bstride, float* loss, float loss_scale, int l2norm, hipStream_t s) {
    const int Tq = Wd / ov, M = B * Tq;
    if (Np % 16 || Kq % 16 || Cp % 16 || bk % 16 || Kq % bk || dt % 4 || G < 1 || G > PVQF_GMAX || G * dt > Np || splits < 1 || splits > PVQF_WAVES || Ksz < 1) return -1;
    const int kIters = Kq / bk, per = (kIters + splits - 1) / splits;      // gemm_engine.h launch_tile: the slice boundaries of the engine's split-K
    if ((kIters + per - 1) / per != splits) return -1;
    PvqFusedArgs a{enc, dec, wd, cbn, c2, cbraw, wup, tab, gq, out, codes, bstride, loss, loss_scale, M, Tq, Hq, Wd, Cp, ov, Kq, per * bk, splits, G, Ksz, d, l2norm, nullptr};
#ifdef ESCX_PVQ_TRACE
    {   // slot n of the buffer for the n-th fused launch since the buffer was (re)set: 4096 workgroups x 8 stamps per slot
        static unsigned long long* last = nullptr; static int n = 0;
        unsigned long long* p = debug_trace_buffer();
        if (p != last) { last = p; n = 0; }
        a.trace = p ? p + (size_t)(n++ % 16) * 4096 * 8 : nullptr;
    }
#endif
    const int NT = Np / 16, STEPS = dt / 4;
    if (NT == 6 && STEPS == 8) { launch_pvq_fused<6, 8>(a, s); return 0; }
    if (NT == 3 && STEPS == 4) { launch_pvq_fused<3, 4>(a, s); return 0; }
    if (NT == 3 && STEPS == 3) { launch_pvq_fused<3, 3>(a, s); return 0; }
    if (NT == 2 && STEPS == 2) { launch_pvq_fused<2, 2>(a, s); return 0; }
    if (NT == 1 && STEPS == 1) { launch_pvq_fused<1, 1>(a, s); return 0; }
    return -1;
}

int pvq_up(const long long* codes, long long bstride, const float* cbraw, int G, int Ksz, int dt, int B, int Hq, int Wd, int Cp, int ov,
           const float* W, int Np, int Kp, const float* dec, float* out, hipStream_t s) {
    const int Tq = Wd / ov, M = B * Tq;
    if (Np % 16 || Kp % 16 || Cp % 16 || dt % 4) return -1;
    PvqUpArgs a{codes, bstride, cbraw, W, dec, out, G, Ksz, dt, Tq, M, Hq, Wd, Cp, ov, Kp, Np / 16, 0};
    const int mt = (M + 15) / 16;
    // enough workgroups for the chip (the launch is HBM-bound: dec read + out written), whole multiples of the 16 tiles a workgroup has in flight
    int per = 16;
    while (per < a.NT && (long long)mt * ((a.NT + per - 1) / per) > 2048) per += 16;
    a.nt_per_wg = per;
    const dim3 grid(mt, (a.NT + per - 1) / per);
    switch (Kp / 16) {
        case 1: ESCX_LAUNCH(pvq_up_kernel<1>, grid, dim3(256), 0, s, a); return 0;
        case 2: ESCX_LAUNCH(pvq_up_kernel<2>, grid, dim3(256), 0, s, a); return 0;
        case 3: ESCX_LAUNCH(pvq_up_kernel<3>, grid, dim3(256), 0, s, a); return 0;
        case 4: ESCX_LAUNCH(pvq_up_kernel<4>, grid, dim3(256), 0, s, a); return 0;
        case 6: ESCX_LAUNCH(pvq_up_kernel<6>, grid, dim3(256), 0, s, a); return 0;
        default: return -1;
    }
}

void istft_ola(const float* frames, const float* win2, float* wave, int B, int T, int ldf, int win, int hop, int left, int half,
               int out_len, hipStream_t s) {
    const long long n = (long long)B * out_len;
    ESCX_LAUNCH(istft_ola_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, frames, win2, wave, B, T, ldf, win, hop,
                       left, half, out_len);
}

void pad_rows(const float* src, float* dst, long long rows, int C, int Cp, hipStream_t s) {
    const long long n = rows * Cp;
    ESCX_LAUNCH(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, C, Cp);
}
void unpad_rows(const float* src, float* dst, long long rows, int C, int Cp, hipStream_t s) {
    const long long n = rows * C;
    ESCX_LAUNCH(unpad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, C, Cp);
}
int deembed_border(const float* x, const float* wv, const float* bv, float* out, int B, int H, int W, int C, int Cp, int pf, int pt,
                   int in_dim, int Fp, hipStream_t s) {
    const int per_clip = 2 * W + 2 * (H > 2 ? H - 2 : 0);
    const long long waves = (long long)B * per_clip;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    const int NO = in_dim * pf * pt;
    if (NO == 12) ESCX_LAUNCH((deembed_border_kernel<12>), dim3(grid), dim3(256), 0, s, x, wv, bv, out, B, H, W, C, Cp, pf, pt, in_dim, Fp);
    else if (NO == 8) ESCX_LAUNCH((deembed_border_kernel<8>), dim3(grid), dim3(256), 0, s, x, wv, bv, out, B, H, W, C, Cp, pf, pt, in_dim, Fp);
    else if (NO == 6) ESCX_LAUNCH((deembed_border_kernel<6>), dim3(grid), dim3(256), 0, s, x, wv, bv, out, B, H, W, C, Cp, pf, pt, in_dim, Fp);
    else if (NO == 4) ESCX_LAUNCH((deembed_border_kernel<4>), dim3(grid), dim3(256), 0, s, x, wv, bv, out, B, H, W, C, Cp, pf, pt, in_dim, Fp);
    else return -1;
    return 0;
}

void codes_pack10(const long long* in, unsigned char* out, long long n, hipStream_t s) {
    const long long q = (n + 3) / 4;
    ESCX_LAUNCH(codes_pack10_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, s, in, out, n);
}
void codes_unpack10(const unsigned char* in, long long* out, long long n, hipStream_t s) {
    const long long q = (n + 3) / 4;
    ESCX_LAUNCH(codes_unpack10_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, s, in, out, n);
}
void test_math(const float* x, float* y, long long n, int which, hipStream_t s) {
    ESCX_LAUNCH(test_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n, which);
}
void test_copy_rows(const float* src, float* dst, long long rows, int Cp, hipStream_t s) {
    ESCX_LAUNCH(test_copy_rows_kernel, dim3(4096), dim3(256), 0, s, src, dst, rows, Cp);
}
void codes_narrow(const long long* in, short* out, long long n, hipStream_t s) {
    ESCX_LAUNCH(codes_narrow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}
void codes_widen(const short* in, long long* out, long long n, hipStream_t s) {
    ESCX_LAUNCH(codes_widen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

}  // namespace escx
