// libescx host side, part 2 of 3: per-stream scratch, launch error check and the per-launch profiler (escx_profile_enable / escx_profile_report; ProfScope in escx_internal.h,
// dispatch-timed launches in launch_prof.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

#include "escx_internal.h"

using namespace escx;

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

