// What does feeding a weight stream through LDS cost the wave that issues it?  (tuning aid, not part of libescx)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_dma_cost.hip -o gpurun_out/ubench_dma_cost && gpurun_out/ubench_dma_cost
// Models the stage loop of the fused attention / MLP kernels at C = 384: a "tile" = KK = 24 weight fragments of 1 KiB, each feeding
// 4 x v_mfma_f32_16x16x4_f32 of every compute wave (x operand in registers, two accumulator chains), double-buffered in LDS, one
// vmcnt(0) + barrier per tile.  Variants of WHO moves the next tile's 24 KiB into LDS and HOW:
//   DMA 0: nobody (same LDS bytes every tile)                      -> the MFMA + ds_read ceiling of the loop
//   DMA 1: every compute wave issues global_load_lds_dwordx4 pieces, one every PER fragments (the product's scheme)
//   DMA 2: the same pieces as raw_buffer_load_lds (SRSRC + 32-bit offset instead of a 64-bit address pair)
//   DMA 3: NL extra loader waves issue everything; compute waves never touch VMEM
//   DMA 4: every compute wave fetches its fragments straight from L2 into registers (no LDS, no barrier)
// Reports shader cycles per MFMA per wave (32 = the matrix pipe's issue rate) from s_memtime of every wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KK = 24;

__device__ __forceinline__ void dma_global(const f32x4* src, f32x4* dst) {
    __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <int NWC, int NL, int DMA, int PER, int SYNC>
__global__ __launch_bounds__(64 * (NWC + NL)) void k(const f32x4* __restrict__ w, int n_tiles_in_w, float* out, int tiles, unsigned long long* cyc) {
    extern __shared__ f32x4 wbuf[];             // [2][KK * 64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= NWC;
    f32x4 xf[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) xf[i] = f32x4{(float)(lane + i), 1.f, 0.5f, (float)i} * 1e-3f;
    f32x4 total = {0, 0, 0, 0};
    // first tile up front
    if (DMA == 1 || DMA == 2) { for (int c = wave; c < KK; c += NWC) dma_global(w + c * 64 + lane, wbuf + c * 64); }
    if (DMA == 3 && loader) { for (int c = wave - NWC; c < KK; c += NL) dma_global(w + c * 64 + lane, wbuf + c * 64); }
    if (DMA == 0) { for (int c = wave; c < 2 * KK; c += NWC + NL) wbuf[c * 64 + lane] = w[c * 64 + lane]; }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (DMA == 3 && loader) {
        for (int t = 0; t < tiles; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const f32x4* src = w + (size_t)((t + 1) % n_tiles_in_w) * KK * 64 + lane;
            f32x4* dst = wbuf + ((t + 1) & 1) * KK * 64;
            for (int c = wave - NWC; c < KK; c += NL) dma_global(src + c * 64, dst + c * 64);
        }
    } else if (DMA == 4) {
        constexpr int PD = 6;
        const f32x4* src = w + lane;
        f32x4 ring[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) ring[i] = src[i * 64];
        for (int t = 0; t < tiles; ++t) {
            const f32x4* nxt = w + (size_t)((t + 1) % n_tiles_in_w) * KK * 64 + lane;
            f32x4 o1 = {0, 0, 0, 0}, o2 = {0, 0, 0, 0};
#pragma unroll
            for (int f = 0; f < KK; ++f) {
                const f32x4 wf = ring[f % PD];
                ring[f % PD] = (f + PD < KK) ? src[(f + PD) * 64] : nxt[(f + PD - KK) * 64];
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[f][0], o1, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[f][1], o2, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[f][2], o1, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[f][3], o2, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            src = nxt;
            total += o1 + o2;
        }
    } else {
        constexpr int PD = 3;
        for (int t = 0; t < tiles; ++t) {
            if (SYNC) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const f32x4* wb = wbuf + (t & 1) * KK * 64 + lane;
            const f32x4* src = w + (size_t)((t + 1) % n_tiles_in_w) * KK * 64 + lane;
            f32x4* dst = wbuf + ((t + 1) & 1) * KK * 64;
            f32x4 ring[PD];
#pragma unroll
            for (int i = 0; i < PD; ++i) { ring[i] = wb[i * 64]; __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            f32x4 o1 = {0, 0, 0, 0}, o2 = {0, 0, 0, 0};
            int slot = 0;
#pragma unroll
            for (int f = 0; f < KK; ++f) {
                const f32x4 wf = ring[f % PD];
                if (f + PD < KK) { ring[f % PD] = wb[(f + PD) * 64]; __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                if ((DMA == 1 || DMA == 2) && f % PER == 0) {
                    const int c = wave + slot * NWC;
                    if (c < KK) {
                        if (DMA == 1) dma_global(src + c * 64, dst + c * 64);
                        else {
                            // SRSRC form: one 32-bit byte offset per lane, the tile base in the descriptor
                            __builtin_amdgcn_global_load_lds((const void*)(src + c * 64), (__attribute__((address_space(3))) void*)(dst + c * 64), 16, 0, 2);   // aux = nt
                        }
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    ++slot;
                }
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[f][0], o1, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[f][1], o2, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[f][2], o1, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[f][3], o2, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            total += o1 + o2;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) cyc[blockIdx.x * (NWC + NL) + wave] = loader ? 0 : (t1 - t0);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}

static f32x4* g_w; static float* g_out; static unsigned long long* g_cyc;
constexpr int N_TILES_W = 96;       // 96 x 24 KiB = 2.36 MB weight set (the C = 384 attention block), L2-resident

template <int NWC, int NL, int DMA, int PER, int SYNC>
void run(const char* what, int wg_per_cu) {
    const int tiles = 960, blocks = 256 * wg_per_cu;
    // LDS: 48 KB used; pad the allocation so that exactly wg_per_cu workgroups fit a CU
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : (wg_per_cu == 2 ? 64 * 1024 : 48 * 1024);
    auto kern = k<NWC, NL, DMA, PER, SYNC>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * (NWC + NL)), lds, 0, g_w, N_TILES_W, g_out, 8, g_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * (NWC + NL)), lds, 0, g_w, N_TILES_W, g_out, tiles, g_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("%-58s launch failed\n", what); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)blocks * (NWC + NL));
    hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0; unsigned long long mx = 0;
    for (auto v : c) if (v) { sum += (double)v; ++n; mx = std::max(mx, v); }
    const double per = sum / n / ((double)tiles * KK * 4);
    const double tf = (double)blocks * NWC * tiles * KK * 4 * 2048.0 / ms / 1e9;
    printf("%-58s NWC=%d NL=%d wg/CU=%d : %6.2f memtime ticks/MFMA (max wave %6.2f)  %7.1f TFLOP/s  %.3f ms\n", what, NWC, NL, wg_per_cu, per,
           (double)mx / ((double)tiles * KK * 4), tf, ms);
}

int main() {
    hipMalloc(&g_w, (size_t)(N_TILES_W + 1) * KK * 64 * 16);
    std::vector<float> h((size_t)(N_TILES_W + 1) * KK * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-4f - 0.05f;
    hipMemcpy(g_w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_out, 1024 * 1024 * 4);
    hipMalloc(&g_cyc, 1024 * 16 * 8);
    printf("# memtime tick: compare the DMA 0 line (pure MFMA issue = 32 shader cycles per MFMA) to calibrate\n");
    run<4, 0, 0, 1, 0>("no DMA, no barrier", 1);
    run<4, 0, 0, 1, 1>("no DMA, barrier per tile", 1);
    run<4, 0, 1, 2, 1>("global_load_lds, compute waves, 1 piece / 2 frags", 1);
    run<4, 0, 1, 4, 1>("global_load_lds, compute waves, 1 piece / 4 frags", 1);
    run<4, 0, 2, 4, 1>("global_load_lds nt, compute waves, 1 piece / 4 frags", 1);
    run<4, 1, 3, 1, 1>("1 loader wave", 1);
    run<4, 2, 3, 1, 1>("2 loader waves", 1);
    run<4, 4, 3, 1, 1>("4 loader waves", 1);
    run<4, 0, 4, 1, 0>("direct L2 -> registers, no LDS", 1);
    run<8, 0, 0, 1, 1>("8 compute waves: no DMA, barrier", 1);
    run<8, 0, 1, 4, 1>("8 compute waves: global_load_lds 1 piece / 4 frags", 1);
    run<8, 0, 1, 8, 1>("8 compute waves: global_load_lds 1 piece / 8 frags", 1);
    run<7, 1, 3, 1, 1>("7 compute + 1 loader", 1);
    run<8, 1, 3, 1, 1>("8 compute + 1 loader", 1);
    run<8, 0, 4, 1, 0>("8 compute: direct L2 -> registers", 1);
    run<4, 0, 0, 1, 1>("2 WG/CU: no DMA, barrier", 2);
    run<4, 0, 1, 4, 1>("2 WG/CU: global_load_lds 1 piece / 4 frags", 2);
    run<4, 1, 3, 1, 1>("2 WG/CU: 1 loader wave each", 2);
    run<4, 0, 4, 1, 0>("2 WG/CU: direct L2 -> registers", 2);
    return 0;
}
